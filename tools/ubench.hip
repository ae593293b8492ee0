// ubench.hip -- per-instruction issue cost on gfx950 for the fp64 mix the objective kernel
// uses (the public guides list fp32/MFMA costs only).  Each kernel runs ITER x 16 copies of
// one instruction over 8 independent register chains; blocks of 256 threads put one wave on
// each SIMD of a CU, and k blocks per CU give k waves per SIMD.
// Output: ns per wave-instruction per SIMD, and the same in cycles at the measured clock
// (s_memtime / s_memrealtime ratio).
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e));          \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

constexpr int ITER = 4096;

#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

#define KERNEL_D3(NAME, INSTR)                                                                      \
    __global__ void NAME(double *out, double a, double b)                                           \
    {                                                                                               \
        double r[8];                                                                                \
        for (int i = 0; i < 8; ++i) r[i] = a + threadIdx.x * 1e-9 + i;                              \
        for (int it = 0; it < ITER; ++it) {                                                         \
            _Pragma("unroll") for (int h = 0; h < 2; ++h)                                          \
            {                                                                                       \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(INSTR " %0, %0, %1, %2" \
                                                                           : "+v"(r[i])             \
                                                                           : "v"(a), "v"(b));       \
            }                                                                                       \
        }                                                                                           \
        double s = 0;                                                                               \
        for (int i = 0; i < 8; ++i) s += r[i];                                                      \
        if (s == 12345.678) out[0] = s;                                                             \
    }

#define KERNEL_D2(NAME, INSTR)                                                                   \
    __global__ void NAME(double *out, double a, double b)                                        \
    {                                                                                            \
        double r[8];                                                                             \
        for (int i = 0; i < 8; ++i) r[i] = a + threadIdx.x * 1e-9 + i;                           \
        for (int it = 0; it < ITER; ++it) {                                                      \
            _Pragma("unroll") for (int h = 0; h < 2; ++h)                                       \
            {                                                                                    \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(INSTR " %0, %0, %1"  \
                                                                           : "+v"(r[i])          \
                                                                           : "v"(b));            \
            }                                                                                    \
        }                                                                                        \
        double s = 0;                                                                            \
        for (int i = 0; i < 8; ++i) s += r[i];                                                   \
        if (s == 12345.678) out[0] = s;                                                          \
    }

#define KERNEL_D1(NAME, INSTR)                                                                  \
    __global__ void NAME(double *out, double a, double b)                                       \
    {                                                                                           \
        double r[8];                                                                            \
        for (int i = 0; i < 8; ++i) r[i] = a + threadIdx.x * 1e-9 + i;                          \
        for (int it = 0; it < ITER; ++it) {                                                     \
            _Pragma("unroll") for (int h = 0; h < 2; ++h)                                      \
            {                                                                                   \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(INSTR " %0, %0"     \
                                                                           : "+v"(r[i]));       \
            }                                                                                   \
        }                                                                                       \
        double s = 0;                                                                           \
        for (int i = 0; i < 8; ++i) s += r[i];                                                  \
        if (s == 12345.678) out[0] = s;                                                         \
    }

// f32 forms
#define KERNEL_F3(NAME, INSTR)                                                                      \
    __global__ void NAME(double *out, double a, double b)                                           \
    {                                                                                               \
        float r[8];                                                                                 \
        float fa = (float)a, fb = (float)b;                                                         \
        for (int i = 0; i < 8; ++i) r[i] = fa + threadIdx.x * 1e-6f + i;                            \
        for (int it = 0; it < ITER; ++it) {                                                         \
            _Pragma("unroll") for (int h = 0; h < 2; ++h)                                          \
            {                                                                                       \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(INSTR " %0, %0, %1, %2" \
                                                                           : "+v"(r[i])             \
                                                                           : "v"(fa), "v"(fb));     \
            }                                                                                       \
        }                                                                                           \
        float s = 0;                                                                                \
        for (int i = 0; i < 8; ++i) s += r[i];                                                      \
        if (s == 12345.678f) out[0] = s;                                                            \
    }

#define KERNEL_F1(NAME, INSTR)                                                                 \
    __global__ void NAME(double *out, double a, double b)                                      \
    {                                                                                          \
        float r[8];                                                                            \
        for (int i = 0; i < 8; ++i) r[i] = (float)a + threadIdx.x * 1e-6f + i;                 \
        for (int it = 0; it < ITER; ++it) {                                                    \
            _Pragma("unroll") for (int h = 0; h < 2; ++h)                                     \
            {                                                                                  \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(INSTR " %0, %0"    \
                                                                           : "+v"(r[i]));      \
            }                                                                                  \
        }                                                                                      \
        float s = 0;                                                                           \
        for (int i = 0; i < 8; ++i) s += r[i];                                                 \
        if (s == 12345.678f) out[0] = s;                                                       \
    }

// packed f32: 64-bit register pairs
#define KERNEL_PK(NAME, INSTR) KERNEL_D3(NAME, INSTR)

// conversions: f64 -> f32 -> f64 round trip counted as two instructions
__global__ void k_cvt_roundtrip(double *out, double a, double b)
{
    double r[8];
    for (int i = 0; i < 8; ++i) r[i] = a + threadIdx.x * 1e-9 + i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float t;
            asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(t) : "v"(r[i]));
            asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(r[i]) : "v"(t));
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += r[i];
    if (s == 12345.678) out[0] = s;
}

__global__ void k_cvt_f32_f64(double *out, double a, double b)
{
    double r[8];
    float t[8];
    for (int i = 0; i < 8; ++i) r[i] = a + threadIdx.x * 1e-9 + i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(t[i]) : "v"(r[i]));
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += t[i];
    if (s == 12345.678f) out[0] = s;
}

__global__ void k_cvt_f64_f32(double *out, double a, double b)
{
    double r[8];
    float t[8];
    for (int i = 0; i < 8; ++i) t[i] = (float)a + threadIdx.x * 1e-6f + i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(r[i]) : "v"(t[i]));
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += r[i];
    if (s == 12345.678) out[0] = s;
}

__global__ void k_cvt_i32_f64(double *out, double a, double b)
{
    double r[8];
    int t[8];
    for (int i = 0; i < 8; ++i) r[i] = a + threadIdx.x * 1e-9 + i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(t[i]) : "v"(r[i]));
    }
    int s = 0;
    for (int i = 0; i < 8; ++i) s += t[i];
    if (s == 123456789) out[0] = s;
}

__global__ void k_ldexp_f64(double *out, double a, double b)
{
    double r[8];
    int e = (int)b;
    for (int i = 0; i < 8; ++i) r[i] = a + threadIdx.x * 1e-9 + i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(r[i]) : "v"(e));
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += r[i];
    if (s == 12345.678) out[0] = s;
}

__global__ void k_cmp_f64(double *out, double a, double b)
{
    double r[8];
    for (int i = 0; i < 8; ++i) r[i] = a + threadIdx.x * 1e-9 + i;
    unsigned long long acc = 0;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                unsigned long long m;
                asm volatile("v_cmp_lt_f64 %0, %1, %2" : "=s"(m) : "v"(r[i]), "v"(b));
                acc ^= m;
            }
    }
    if (acc == 12345) out[0] = 1;
}

// broadcast LDS read of 48 B (3 x ds_read_b128) per "instruction group"
__global__ void k_lds_bcast(double *out, double a, double b)
{
    __shared__ double4 lds[256];
    lds[threadIdx.x] = make_double4(a, b, a, b);
    __syncthreads();
    double s = 0;
    int idx = (int)b & 63;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double4 v = lds[(idx + i + it) & 255];
            asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
            s += v.x;
        }
    }
    if (s == 12345.678) out[0] = s;
}


// does the quarter-rate v_rcp_f64 overlap with full-rate fp64 FMAs of the same / other waves?
// 4 rcp + 12 fma per iteration: 4*16 + 12*4 = 112 cycles if serial, ~64 if they overlap.
__global__ void k_mix_rcp_fma(double *out, double a, double b)
{
    double r[8], t[4];
    for (int i = 0; i < 8; ++i) r[i] = a + threadIdx.x * 1e-9 + i;
    for (int i = 0; i < 4; ++i) t[i] = b + i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            asm volatile("v_rcp_f64 %0, %0" : "+v"(t[i]));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r[2 * i]) : "v"(a), "v"(b));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r[2 * i + 1]) : "v"(a), "v"(b));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r[(2 * i + 2) & 7]) : "v"(a), "v"(b));
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += r[i];
    for (int i = 0; i < 4; ++i) s += t[i];
    if (s == 12345.678) out[0] = s;
}

// clock measurement: ratio of s_memtime (shader clock) to s_memrealtime (100 MHz)
__global__ void k_clock(unsigned long long *out)
{
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    double x = threadIdx.x;
    for (int i = 0; i < 2000000; ++i) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(x));
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = t1 - t0;
        out[1] = r1 - r0;
    }
    if (x == 1.2345) out[2] = 1;
}

KERNEL_D3(k_fma_f64, "v_fma_f64")
KERNEL_D2(k_mul_f64, "v_mul_f64")
KERNEL_D2(k_add_f64, "v_add_f64")
KERNEL_D2(k_min_f64, "v_min_f64")
KERNEL_D1(k_rcp_f64, "v_rcp_f64")
KERNEL_D1(k_rsq_f64, "v_rsq_f64")
KERNEL_D1(k_sqrt_f64, "v_sqrt_f64")
KERNEL_D1(k_rndne_f64, "v_rndne_f64")
KERNEL_D1(k_fract_f64, "v_fract_f64")
KERNEL_F3(k_fma_f32, "v_fma_f32")
KERNEL_PK(k_pk_fma_f32, "v_pk_fma_f32")
KERNEL_F1(k_rcp_f32, "v_rcp_f32")
KERNEL_F1(k_exp_f32, "v_exp_f32")
KERNEL_F1(k_mov_b32, "v_mov_b32")

typedef void (*kern_t)(double *, double, double);

struct Case {
    const char *name;
    kern_t fn;
    double instr_per_iter;
};

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device: %s  arch %s  CUs %d  clockRate %d kHz\n", prop.name, prop.gcnArchName, cus, prop.clockRate);
    double *d_out;
    CHECK(hipMalloc(&d_out, 64));
    unsigned long long *d_clk, h_clk[2];
    CHECK(hipMalloc(&d_clk, 64));
    // warm up + clock under fp64 load on every CU
    hipLaunchKernelGGL(k_clock, dim3(cus * 4), dim3(256), 0, 0, d_clk);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_clock, dim3(cus * 4), dim3(256), 0, 0, d_clk);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h_clk, d_clk, sizeof h_clk, hipMemcpyDeviceToHost));
    const double ghz = (double)h_clk[0] / ((double)h_clk[1] * 10.0);   // ticks per 10 ns
    printf("shader clock under all-CU fp64 fma load: %.3f GHz (s_memtime %llu / s_memrealtime %llu)\n", ghz,
           h_clk[0], h_clk[1]);

    std::vector<Case> cases = {
        {"v_fma_f64", k_fma_f64, 16},     {"v_mul_f64", k_mul_f64, 16},       {"v_add_f64", k_add_f64, 16},
        {"v_min_f64", k_min_f64, 16},     {"v_rcp_f64", k_rcp_f64, 16},       {"v_rsq_f64", k_rsq_f64, 16},
        {"v_sqrt_f64", k_sqrt_f64, 16},   {"v_rndne_f64", k_rndne_f64, 16},   {"v_fract_f64", k_fract_f64, 16},
        {"v_ldexp_f64", k_ldexp_f64, 16}, {"v_cmp_lt_f64", k_cmp_f64, 16},    {"v_cvt_f32_f64", k_cvt_f32_f64, 16},
        {"v_cvt_f64_f32", k_cvt_f64_f32, 16}, {"v_cvt_i32_f64", k_cvt_i32_f64, 16},
        {"cvt f64->f32->f64 (2 instr)", k_cvt_roundtrip, 16},
        {"v_fma_f32", k_fma_f32, 16},     {"v_pk_fma_f32", k_pk_fma_f32, 16}, {"v_rcp_f32", k_rcp_f32, 16},
        {"v_exp_f32", k_exp_f32, 16},     {"v_mov_b32", k_mov_b32, 16},       {"3x ds_read_b128 bcast", k_lds_bcast, 16},
        {"mix 4 rcp_f64 + 12 fma_f64 /16", k_mix_rcp_fma, 16},
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%-30s", "instruction  \\  waves/SIMD");
    const int occ[] = {1, 2, 4, 8};
    for (int k : occ) printf("   k=%d ns (cyc)  ", k);
    printf("\n");
    for (const Case &c : cases) {
        printf("%-30s", c.name);
        for (int k : occ) {
            hipLaunchKernelGGL(c.fn, dim3(cus * k), dim3(256), 0, 0, d_out, 1.000001, 0.999999);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, 0));
            const int reps = 3;
            for (int r = 0; r < reps; ++r)
                hipLaunchKernelGGL(c.fn, dim3(cus * k), dim3(256), 0, 0, d_out, 1.000001, 0.999999);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            // per SIMD: k waves x ITER x instr_per_iter wave-instructions in ms/reps
            const double n = (double)k * ITER * c.instr_per_iter;
            const double ns = (double)ms / reps * 1e6 / n;
            printf("  %6.2f (%5.2f)   ", ns, ns * ghz);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
