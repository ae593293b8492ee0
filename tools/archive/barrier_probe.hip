// barrier_probe.hip -- what does one grid-wide exchange cost inside a persistent kernel on MI355X?
//
//   hipcc --offload-arch=gfx950 -O2 tools/barrier_probe.hip -o tools/barrier_probe && tools/barrier_probe
//
// Every workgroup posts a record of R doubles, all workgroups meet at a barrier, every workgroup
// reads all posts and folds them -- the communication pattern of one swarm generation when the
// swarm state stays with its owner workgroup for the whole run.  Two ways of publishing the posts:
//   fence : plain stores + __threadfence() (agent-scope release: an L2 write-back on a multi-XCD part)
//   atomic: agent-scope relaxed atomic stores / loads (write-through, no L2 write-back) + s_waitcnt
// Launched with hipLaunchCooperativeKernel (the runtime refuses a grid that cannot be co-resident)
// and every spin loop gives up after ~0.2 s of s_memrealtime, so a mistake cannot hang the GPU.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e = (x);                                                           \
        if (e != hipSuccess) {                                                        \
            printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__);      \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

struct Args {
    unsigned long long *count;   // monotonically increasing arrival counter
    double *posts;               // [nwg][R]
    double *out;                 // [nwg] checksum per workgroup
    int *err;
    int iters, R, mode;
};

__device__ __forceinline__ bool barrier(unsigned long long *count, unsigned long long target, int *err)
{
    __syncthreads();
    __shared__ int ok;
    if (threadIdx.x == 0) {
        ok = 1;
        __hip_atomic_fetch_add(count, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 20000000ull) {   // 0.2 s at 100 MHz
                ok = 0;
                *err = 1;
                break;
            }
        }
    }
    __syncthreads();
    return ok != 0;
}

__global__ __launch_bounds__(256) void probe(Args a)
{
    const unsigned nwg = gridDim.x;
    double acc = 0.0;
    for (int it = 0; it < a.iters; ++it) {
        // post (two alternating slots so a fast workgroup cannot overwrite what a slow one still reads)
        double *mine = a.posts + ((size_t)(it & 1) * nwg + blockIdx.x) * a.R;
        for (int r = threadIdx.x; r < a.R; r += blockDim.x) {
            const double v = (double)(blockIdx.x + 1) * (it + 1) + r;
            if (a.mode == 0)
                mine[r] = v;
            else
                __hip_atomic_store(mine + r, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (a.mode == 0)
            __threadfence();
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the write-through stores are done (a workgroup-scope
                                                               // fence or a barrier does NOT wait for them on gfx950)
        if (!barrier(a.count, (unsigned long long)(it + 1) * nwg, a.err)) return;
        if (a.mode == 0) __threadfence();
        // fold: every workgroup reads every post
        const double *all = a.posts + (size_t)(it & 1) * nwg * a.R;
        double s = 0.0;
        for (unsigned i = threadIdx.x; i < nwg * (unsigned)a.R; i += blockDim.x) {
            if (a.mode == 0)
                s += ((const volatile double *)all)[i];
            else
                s += __hip_atomic_load(all + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        acc += s;
    }
    // block-reduce acc (order does not matter for the check: integers)
    __shared__ double red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.out[blockIdx.x] = red[0];
}

int main()
{
    const int iters = 2000;
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    printf("device has %d CUs\n", cus);
    for (int mode = 0; mode < 2; ++mode)
        for (int R : {2, 24, 78})
            for (int nwg : {8, 51, 204, 256, 512, 768}) {
                Args a;
                a.iters = iters;
                a.R = R;
                a.mode = mode;
                CHECK(hipMalloc(&a.count, 8));
                CHECK(hipMemset(a.count, 0, 8));
                CHECK(hipMalloc(&a.posts, sizeof(double) * 2 * nwg * R));
                CHECK(hipMalloc(&a.out, sizeof(double) * nwg));
                CHECK(hipMalloc(&a.err, 4));
                CHECK(hipMemset(a.err, 0, 4));
                void *params[] = {&a};
                hipEvent_t e0, e1;
                CHECK(hipEventCreate(&e0));
                CHECK(hipEventCreate(&e1));
                CHECK(hipEventRecord(e0, 0));
                hipError_t le = hipLaunchCooperativeKernel((void *)probe, dim3(nwg), dim3(256), params, 0, 0);
                if (le != hipSuccess) {
                    printf("mode %d R %3d nwg %4d: cooperative launch refused (%s)\n", mode, R, nwg, hipGetErrorString(le));
                    (void)hipGetLastError();
                    continue;
                }
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                int err = 0;
                CHECK(hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost));
                std::vector<double> out(nwg);
                CHECK(hipMemcpy(out.data(), a.out, sizeof(double) * nwg, hipMemcpyDeviceToHost));
                // expected: sum over it, wg, r of (wg+1)*(it+1) + r
                double expect = 0.0;
                for (int it = 0; it < iters; ++it)
                    for (int w = 0; w < nwg; ++w)
                        for (int r = 0; r < R; ++r) expect += (double)(w + 1) * (it + 1) + r;
                bool good = !err;
                for (int w = 0; w < nwg; ++w) good = good && out[w] == expect;
                printf("%-6s R %3d nwg %4d: %7.2f us per exchange  %s\n", mode ? "atomic" : "fence", R, nwg,
                       ms * 1e3 / iters, good ? "ok" : (err ? "TIMEOUT" : "WRONG SUMS"));
                CHECK(hipFree(a.count));
                CHECK(hipFree(a.posts));
                CHECK(hipFree(a.out));
                CHECK(hipFree(a.err));
            }
    return 0;
}
