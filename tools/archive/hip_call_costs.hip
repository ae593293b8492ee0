// What the HIP calls of a context's life cost on the host (one MI355X box): the pieces of nmrfit_ctx_create /
// nmrfit_ctx_destroy, timed one by one, five rounds.   hipcc --offload-arch=gfx950 -O2 hip_call_costs.hip -o hip_call_costs
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void tiny(double *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipSetDevice(0);
    hipFree(nullptr);
    std::vector<double> host(4096, 1.0);
    for (int round = 0; round < 5; ++round) {
        double t0 = now();
        hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        double t1 = now();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        double t2 = now();
        double *d = nullptr; hipMalloc((void **)&d, 5 * 4096 * 8 + 4096);
        double t3 = now();
        hipMemsetAsync(d, 0, 4 * 4096 * 8, st);
        double t4 = now();
        for (int a = 0; a < 4; ++a) hipMemcpyAsync(d + 4 * 4096, host.data(), 4096 * 8, hipMemcpyHostToDevice, st);
        double t5 = now();
        for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(tiny, dim3(16), dim3(256), 0, st, d, 4096);
        double t6 = now();
        hipStreamSynchronize(st);
        double t7 = now();
        double *d2 = nullptr; hipMalloc((void **)&d2, 204 * 22 * 8 * 8);
        double t8 = now();
        hipStreamSynchronize(st);
        hipFree(d2);
        double t9 = now();
        hipFree(d);
        double t10 = now();
        hipEventDestroy(e0); hipEventDestroy(e1);
        double t11 = now();
        hipStreamDestroy(st);
        double t12 = now();
        printf("stream create %.0f us, 2 events %.0f, malloc %.0f, memset %.0f, 4 H2D %.0f, 5 launches %.0f, sync %.0f | 2nd malloc %.0f, free small %.0f, free %.0f, "
               "events destroy %.0f, stream destroy %.0f\n", t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6, t8 - t7, t9 - t8, t10 - t9, t11 - t10, t12 - t11);
    }
    return 0;
}
