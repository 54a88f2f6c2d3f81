// Micro-benchmark: issue cost (cycles per wave-instruction per SIMD) of the VALU ops the Viterbi kernel uses.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define REP16(x) x x x x x x x x x x x x x x x x
#define KERNEL(NAME, BODY)                                                              \
    __global__ __launch_bounds__(256) void NAME(double *out, int iters) {               \
        double a = threadIdx.x * 1.5, b = threadIdx.x * 0.25 + 1, c = 3.0, d = 4.0;       \
        int x = threadIdx.x, y = threadIdx.x * 3, z = 5, w = 7;                           \
        for (int i = 0; i < iters; i++) { REP16(BODY) }                                   \
        out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + x + y + z + w;              \
    }
KERNEL(k_add_f64, asm volatile("v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %3" : "+v"(a), "+v"(b) : "v"(c), "v"(d));)
KERNEL(k_min_f64, asm volatile("v_min_f64 %0, %0, %2\n v_min_f64 %1, %1, %3" : "+v"(a), "+v"(b) : "v"(c), "v"(d));)
KERNEL(k_cmp_f64, asm volatile("v_cmp_lt_f64 vcc, %0, %2\n v_cmp_lt_f64 vcc, %1, %3" : "+v"(a), "+v"(b) : "v"(c), "v"(d) : "vcc");)
KERNEL(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %3, vcc" : "+v"(x), "+v"(y) : "v"(z), "v"(w) : "vcc");)
KERNEL(k_cndmask_sgpr, asm volatile("v_cndmask_b32 %0, %0, %2, s[20:21]\n v_cndmask_b32 %1, %1, %3, s[20:21]" : "+v"(x), "+v"(y) : "v"(z), "v"(w) : "s20", "s21");)
KERNEL(k_cndmask_indep, asm volatile("v_cndmask_b32 %0, %2, %3, vcc\n v_cndmask_b32 %1, %3, %2, vcc" : "=v"(x), "=v"(y) : "v"(z), "v"(w) : "vcc");)
KERNEL(k_cndmask_init, asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %3, vcc" : "+v"(x), "+v"(y) : "v"(z + 1), "v"(w + 1));)
KERNEL(k_cmp_cnd_vcc, asm volatile("v_cmp_eq_u32 vcc, %0, %2\n v_cndmask_b32 %1, %1, %3, vcc" : "+v"(x), "+v"(y) : "v"(z), "v"(w) : "vcc");)
KERNEL(k_cmp_cnd_sgpr, asm volatile("v_cmp_eq_u32 s[20:21], %0, %2\n v_cndmask_b32 %1, %1, %3, s[20:21]" : "+v"(x), "+v"(y) : "v"(z), "v"(w) : "s20", "s21");)
KERNEL(k_cmp64_cnd_vcc, asm volatile("v_cmp_lt_f64 vcc, %0, %2\n s_nop 1\n v_cndmask_b32 %1, %1, %3, vcc" : "+v"(a), "+v"(y) : "v"(c), "v"(w) : "vcc");)
KERNEL(k_cmp64_cnd_sgpr, asm volatile("v_cmp_lt_f64 s[20:21], %0, %2\n s_nop 1\n v_cndmask_b32 %1, %1, %3, s[20:21]" : "+v"(a), "+v"(y) : "v"(c), "v"(w) : "s20", "s21");)
KERNEL(k_mov, asm volatile("v_mov_b32 %0, %2\n v_mov_b32 %1, %3" : "+v"(x), "+v"(y) : "v"(z), "v"(w));)
KERNEL(k_add_u32, asm volatile("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %3" : "+v"(x), "+v"(y) : "v"(z), "v"(w));)
KERNEL(k_dpp, asm volatile("v_mov_b32_dpp %0, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %3 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(x), "+v"(y) : "v"(z), "v"(w));)
KERNEL(k_swap32, asm volatile("v_permlane32_swap_b32 %0, %2\n v_permlane32_swap_b32 %1, %3" : "+v"(x), "+v"(y), "+v"(z), "+v"(w));)
KERNEL(k_swap16, asm volatile("v_permlane16_swap_b32 %0, %2\n v_permlane16_swap_b32 %1, %3" : "+v"(x), "+v"(y), "+v"(z), "+v"(w));)
KERNEL(k_readlane, asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5" : "+v"(x), "+v"(y) :: "s20", "s21");)
KERNEL(k_writelane, asm volatile("v_writelane_b32 %0, s20, 3\n v_writelane_b32 %1, s21, 5" : "+v"(x), "+v"(y) :: "s20", "s21");)
KERNEL(k_cmp_u32, asm volatile("v_cmp_eq_u32 vcc, %0, %2\n v_cmp_eq_u32 vcc, %1, %3" : "+v"(x), "+v"(y) : "v"(z), "v"(w) : "vcc");)
KERNEL(k_lshr64, asm volatile("v_lshrrev_b64 %0, %2, %0\n v_lshrrev_b64 %1, %3, %1" : "+v"(a), "+v"(b) : "v"(z), "v"(w));)
KERNEL(k_fma_f64, asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %3, %2" : "+v"(a), "+v"(b) : "v"(c), "v"(d));)
KERNEL(k_exp_pipe, a = exp(a * 1e-9) + c; b = log(b + 2.0) + d;)

int main() {
    double *d_out;
    const int nblocks = 256 * 8, iters = 2000;     // 8 blocks x 4 waves = 8 waves per SIMD
    CHECK(hipMalloc(&d_out, sizeof(double) * nblocks * 256));
    struct { const char *name; void (*k)(double *, int); double per_iter; } ks[] = {
        {"v_add_f64", k_add_f64, 32}, {"v_min_f64", k_min_f64, 32}, {"v_fma_f64", k_fma_f64, 32}, {"v_cmp_lt_f64", k_cmp_f64, 32},
        {"v_cndmask_b32", k_cndmask, 32}, {"v_cndmask_b32 sgpr mask", k_cndmask_sgpr, 32}, {"v_cndmask_b32 independent", k_cndmask_indep, 32}, {"v_cndmask_b32 z+1 operands", k_cndmask_init, 32}, {"cmp_u32->vcc + cndmask vcc (pair)", k_cmp_cnd_vcc, 16}, {"cmp_u32->sgpr + cndmask sgpr (pair)", k_cmp_cnd_sgpr, 16}, {"cmp_f64->vcc,nop,cndmask vcc (pair)", k_cmp64_cnd_vcc, 16}, {"cmp_f64->sgpr,nop,cndmask sgpr (pair)", k_cmp64_cnd_sgpr, 16}, {"v_mov_b32", k_mov, 32}, {"v_add_u32", k_add_u32, 32}, {"v_cmp_eq_u32", k_cmp_u32, 32},
        {"v_mov_b32_dpp", k_dpp, 32}, {"v_permlane32_swap", k_swap32, 32}, {"v_permlane16_swap", k_swap16, 32},
        {"v_readlane_b32", k_readlane, 32}, {"v_writelane_b32", k_writelane, 32}, {"v_lshrrev_b64", k_lshr64, 32},
        {"exp+log f64 pair (ocml)", k_exp_pipe, 16},
    };
    for (auto &e : ks) {
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        hipLaunchKernelGGL(e.k, dim3(nblocks), dim3(256), 0, 0, d_out, 10);
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(e.k, dim3(nblocks), dim3(256), 0, 0, d_out, iters);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        // waves per SIMD = nblocks*4/1024 ; instructions per wave = iters*per_iter
        double wave_instr_per_simd = (double)nblocks * 4 / 1024 * iters * e.per_iter;
        printf("%-26s %8.3f ms  -> %.2f cycles per wave-instruction per SIMD @2.4GHz\n", e.name, ms, ms * 1e-3 * 2.4e9 / wave_instr_per_simd);
    }
    return 0;
}
