// Micro-benchmark: what a DEPENDENT float64 chain costs per instruction on gfx950, at 1 and 2 waves per SIMD, against
// 2 / 4 independent chains interleaved in one wave -- the question behind the BCJR pass: is a wave that is
// "ready but not issued" waiting for its partner's VALU slots or for its own previous result?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define REP8(x) x x x x x x x x
#define KERNEL(NAME, BODY)                                                                \
    __global__ __launch_bounds__(512) void NAME(double *out, int iters) {                 \
        double a = threadIdx.x * 1e-3 + 0.5, b = a + 0.1, c = a + 0.2, d = a + 0.3;           \
        const double m = 0.999, k = 1e-4;                                                   \
        for (int i = 0; i < iters; i++) { REP8(BODY) }                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;                         \
    }
KERNEL(k_chain1, asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %0, %0, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(k));)
KERNEL(k_chain2, asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(k));)
KERNEL(k_chain4, asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(k));)
// one alpha step of the 4-state fast path: two quad-permuted copies of a (4 DPP moves), a multiply and an FMA, all dependent
template <int CTRL>
__device__ __forceinline__ double dppd(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__global__ __launch_bounds__(512) void k_alpha1(double *out, int iters) {
    double a = threadIdx.x * 1e-3 + 0.5;
    const double m = 0.499, k = 0.5;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const double p = dppd<0x88>(a), q = dppd<0xDD>(a);
            a = __builtin_fma(q, k, p * m);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
// two / four independent alpha chains in one wave (the compiler interleaves them)
template <int NC>
__global__ __launch_bounds__(512) void k_alphaN(double *out, int iters) {
    double a[NC];
    for (int c = 0; c < NC; c++) a[c] = threadIdx.x * 1e-3 + 0.5 + 0.01 * c;
    const double m = 0.499, k = 0.5;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const double p = dppd<0x88>(a[c]), q = dppd<0xDD>(a[c]);
                a[c] = __builtin_fma(q, k, p * m);
            }
        }
    }
    double r = 0;
    for (int c = 0; c < NC; c++) r += a[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
    double *d_out;
    CHECK(hipMalloc(&d_out, sizeof(double) * 256 * 512 * 2));
    const int iters = 20000;
    struct { const char *name; void (*k)(double *, int); double per_iter; } ks[] = {
        {"fma_f64, 1 dependent chain", k_chain1, 32}, {"fma_f64, 2 chains interleaved", k_chain2, 32}, {"fma_f64, 4 chains interleaved", k_chain4, 32},
        {"alpha step x1 (4 dpp + mul + fma = 6 instr)", k_alpha1, 48}, {"alpha step x2 interleaved", k_alphaN<2>, 96}, {"alpha step x4 interleaved", k_alphaN<4>, 192},
    };
    for (int wps = 1; wps <= 2; wps++)
        for (auto &e : ks) {
            hipEvent_t a, b;
            CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
            const int threads = 256 * wps;                            // one workgroup per CU: wps waves per SIMD
            hipLaunchKernelGGL(e.k, dim3(256), dim3(threads), 0, 0, d_out, 10);
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(e.k, dim3(256), dim3(threads), 0, 0, d_out, iters);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            printf("%d wave(s)/SIMD  %-46s %8.3f ms -> %.2f ns per instruction per wave; %.2f ns per instruction per SIMD\n", wps, e.name, ms,
                   ms * 1e6 / (iters * e.per_iter), ms * 1e6 / (iters * e.per_iter * wps));
        }
    return 0;
}
