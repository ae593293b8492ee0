// Does v_mfma_f64_16x16x4_f64 (matrix pipe) overlap with fp64 VALU FMAs on the same SIMD?
// Three kernels per occupancy: VALU only (16 fma per iteration), MFMA only (1 per iteration),
// both.  If the pipes overlap, t(both) ~ max(t(valu), t(mfma)); if they serialise, ~ the sum.
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_overlap.hip -o tools/mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int ITER = 4096;

template <int NFMA, int NMFMA>
__global__ void k(double *out, double a, double b)
{
    double r[8];
    for (int i = 0; i < 8; ++i) r[i] = a + threadIdx.x * 1e-9 + i;
    d4 acc = {0, 0, 0, 0};
    double ma = a + threadIdx.x, mb = b;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int m = 0; m < NMFMA; ++m) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NFMA; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r[i & 7]) : "v"(a), "v"(b));
    }
    double s = acc[0] + acc[1] + acc[2] + acc[3];
    for (int i = 0; i < 8; ++i) s += r[i];
    if (s == 12345.678) out[0] = s;
}

template <int NFMA, int NMFMA>
float run(int cus, int kocc, double *d_out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NFMA, NMFMA>), dim3(cus * kocc), dim3(256), 0, 0, d_out, 1.000001, 0.999999);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<NFMA, NMFMA>), dim3(cus * kocc), dim3(256), 0, 0, d_out, 1.000001, 0.999999);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 3 * 1e6f / (kocc * ITER);   // ns per iteration per wave-slot
}

int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    double *d_out; hipMalloc(&d_out, 64);
    printf("ns per iteration per SIMD (iteration = 16 v_fma_f64 and/or 1 v_mfma_f64_16x16x4)\n");
    printf("waves/SIMD   valu16     mfma1   both(16+1)   both(32+1)  valu32   both(16+2) mfma2\n");
    for (int kocc : {1, 2, 3, 4}) {
        float v16 = run<16, 0>(cus, kocc, d_out), m1 = run<0, 1>(cus, kocc, d_out), b = run<16, 1>(cus, kocc, d_out);
        float b32 = run<32, 1>(cus, kocc, d_out), v32 = run<32, 0>(cus, kocc, d_out), b162 = run<16, 2>(cus, kocc, d_out), m2 = run<0, 2>(cus, kocc, d_out);
        printf("%5d     %8.2f  %8.2f  %9.2f   %9.2f  %8.2f  %9.2f %8.2f\n", kocc, v16, m1, b, b32, v32, b162, m2);
    }
    return 0;
}
