// Accuracy of the two reciprocal-based float64 quotients of csrc/cpx_math.h against IEEE division, on gfx950:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I commpy_amd/csrc -I include scripts/micro/div_nr_check.hip -o /tmp/div_nr_check && /tmp/div_nr_check
// prints, per form, the largest error in ulp of the exact quotient and the fraction of results that are not correctly rounded.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include "cpx_math.h"

__device__ __forceinline__ double div_two_steps(double x, double y) {      // the form of rounds 3 / 4a
    double r = __builtin_amdgcn_rcp(y);
    double e = __builtin_fma(-y, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-y, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q = x * r;
    return __builtin_fma(__builtin_fma(-y, q, x), r, q);
}

__device__ __forceinline__ uint64_t rng(uint64_t &s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }

__global__ void check(int n, double *maxulp, unsigned long long *wrong) {
    uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    double m1 = 0, m2 = 0, m0 = 0, mr = 0;
    unsigned long long w1 = 0, w2 = 0, w0 = 0;
    for (int i = 0; i < n; i++) {
        // operands as the kernels see them: x in (0, 2^40), y in (2^-40, 2^700), also x <= y (the ratio row) every other draw
        const double a = __longlong_as_double((rng(s) & 0x000fffffffffffffull) | ((uint64_t)(1023 - 40 + (rng(s) % 80)) << 52));
        const double b = __longlong_as_double((rng(s) & 0x000fffffffffffffull) | ((uint64_t)(1023 - 40 + (rng(s) % 740)) << 52));
        const double x = (i & 1) ? fmin(a, b) : a, y = (i & 1) ? fmax(a, b) : b;
        const double q = x / y, q2 = div_two_steps(x, y), q1 = cpx::div_nr(x, y), q0 = cpx::div_nr0(x, y);
        const double rr = __builtin_amdgcn_rcp(y), re = 1.0 / y;
        const double ulp = fabs(q) * 0x1p-52;
        const double e2 = fabs(q2 - q) / ulp, e1 = fabs(q1 - q) / ulp;
        m2 = fmax(m2, e2); m1 = fmax(m1, e1); m0 = fmax(m0, fabs(q0 - q) / ulp); mr = fmax(mr, fabs(rr - re) / (fabs(re) * 0x1p-52));
        w2 += q2 != q; w1 += q1 != q; w0 += q0 != q;
    }
    atomicMax((unsigned long long *)&maxulp[0], (unsigned long long)__double_as_longlong(m2));
    atomicMax((unsigned long long *)&maxulp[1], (unsigned long long)__double_as_longlong(m1));
    atomicAdd(&wrong[0], w2);
    atomicAdd(&wrong[1], w1);
    atomicMax((unsigned long long *)&maxulp[2], (unsigned long long)__double_as_longlong(m0));
    atomicMax((unsigned long long *)&maxulp[3], (unsigned long long)__double_as_longlong(mr));
    atomicAdd(&wrong[2], w0);
}

int main() {
    double *d_m; unsigned long long *d_w;
    hipMalloc(&d_m, 32); hipMalloc(&d_w, 32);
    hipMemset(d_m, 0, 32); hipMemset(d_w, 0, 32);
    const int n = 2000, blocks = 1024, threads = 256;
    check<<<blocks, threads>>>(n, d_m, d_w);
    double m[4]; unsigned long long w[4];
    hipMemcpy(m, d_m, 32, hipMemcpyDeviceToHost); hipMemcpy(w, d_w, 32, hipMemcpyDeviceToHost);
    const double tot = (double)n * blocks * threads;
    printf("        (two Newton steps + residual): max error %.3f ulp, not correctly rounded %.3e of %.0f\n", m[0], w[0] / tot, tot);
    printf("div_nr  (one Newton step  + residual): max error %.3f ulp, not correctly rounded %.3e of %.0f\n", m[1], w[1] / tot, tot);
    printf("div_nr0 (rcp + residual)                : max error %.3f ulp, not correctly rounded %.3e\n", m[2], w[2] / tot);
    printf("v_rcp_f64 itself against 1 / y         : max error %.3f ulp\n", m[3]);
    return 0;
}
