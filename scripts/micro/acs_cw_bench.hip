// Micro-benchmark (not part of the product): the add-compare-select recursion of a K=7 rate-1/2 Viterbi decoder with one
// CODEWORD per lane and the 64 path metrics updated IN PLACE.  A radix-2 butterfly reads states (2j, 2j+1) and
// writes (j, j+32); storing the results where the inputs were rotates the logical->physical map by one bit per step,
// so six unrolled steps return to the identity and 128 VGPRs hold the metrics (experiments/viterbi_cw.hip double
// buffered them: 256 VGPR + AGPR spills, one wave per SIMD).  Measures how far the straight-line ACS gets from its
// issue bound and how it scales with waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I commpy_amd/csrc -o /tmp/acs_cw_bench scripts/micro/acs_cw_bench.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cpx_math.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int LGS = 6, S = 64;
#ifdef INF
#define INITV __builtin_huge_val()
#else
#define INITV 1e300
#endif
#ifndef SCALE
#define SCALE 16.0
#endif
#ifndef LEN
#define LEN (2060 + 32)
#endif
#ifndef TT
#define TT 1032
#endif
#ifndef ITERS
#define ITERS 3
#endif
constexpr unsigned G0 = 0155u, G1 = 0117u;

constexpr int parity(unsigned v) { return __builtin_popcount(v) & 1; }
constexpr int code(int s, int j) {
    const unsigned p = (unsigned)(((s << 1) & (S - 1)) | j), b = (unsigned)(s >> (LGS - 1));
    const unsigned reg = (b << LGS) | p;
    return (parity(reg & G0) << 1) | parity(reg & G1);
}
constexpr int rotl6(int s, int r) { r %= 6; return ((s << r) | (s >> (6 - r))) & 63; }

__device__ __forceinline__ double vmin(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned shift_in_lt(unsigned acc, double x, double y) {
    unsigned r;
    asm("v_cmp_lt_f64 vcc, %2, %3\n\tv_addc_co_u32 %0, vcc, %1, %1, vcc" : "=v"(r) : "v"(acc), "v"(x), "v"(y) : "vcc");
    return r;
}

// decision + select in one block: d = (y < x); r = d ? y : x; acc = 2*acc + d
__device__ __forceinline__ double acs_sel(unsigned &acc, double x, double y) {
    int rlo, rhi;
    asm("v_cmp_lt_f64 vcc, %4, %3\n\tv_cndmask_b32 %0, %5, %7, vcc\n\tv_cndmask_b32 %1, %6, %8, vcc\n\t"
        "v_addc_co_u32 %2, vcc, %2, %2, vcc"
        : "=&v"(rlo), "=&v"(rhi), "+v"(acc)
        : "v"(x), "v"(y), "v"(__double2loint(x)), "v"(__double2hiint(x)), "v"(__double2loint(y)), "v"(__double2hiint(y))
        : "vcc");
    return __hiloint2double(rhi, rlo);
}

template <int R, bool ARGMIN, bool BM>
__device__ __forceinline__ void step(double (&pm)[S], double r0, double r1, unsigned long long *dec, unsigned char *best) {
    double bmv[4];
    if (BM) {
        r0 = fmin(fmax(r0, -500.0), 500.0);
        r1 = fmin(fmax(r1, -500.0), 500.0);
        const double n0 = cpx::fast_log(exp(r0) + 1.0), n1 = cpx::fast_log(exp(r1) + 1.0);
        const double m00 = n0, m01 = n0 - r0, m10 = n1, m11 = n1 - r1;
        bmv[0] = (0.0 + m00) + m10; bmv[1] = (0.0 + m00) + m11;
        bmv[2] = (0.0 + m01) + m10; bmv[3] = (0.0 + m01) + m11;
    } else {
        bmv[0] = r0; bmv[1] = r1; bmv[2] = r0 + 1.0; bmv[3] = r1 + 1.0;
    }
    unsigned dw0 = 0, dw1 = 0;
#pragma unroll
    for (int j = 0; j < 32; j++) {
        constexpr int dummy = 0; (void)dummy;
        const int x = rotl6(2 * j, R), y = rotl6(2 * j + 1, R);
        const double a = pm[x], b = pm[y];
        const double a0 = a + bmv[code(j, 0)], a1 = b + bmv[code(j, 1)];
        const double b0 = a + bmv[code(j + 32, 0)], b1 = b + bmv[code(j + 32, 1)];
#ifdef SEL
        pm[x] = acs_sel(dw0, a0, a1);
        pm[y] = acs_sel(dw1, b0, b1);
#else
        dw0 = shift_in_lt(dw0, a1, a0);
        dw1 = shift_in_lt(dw1, b1, b0);
        pm[x] = vmin(a0, a1);
        pm[y] = vmin(b0, b1);
#endif
    }
    int bst = 0;
    unsigned long long k0, k1, k2, k3;
    (void)k0; (void)k1; (void)k2; (void)k3;
    if (ARGMIN) {
        double m0 = pm[0], m1 = pm[1], m2 = pm[2], m3 = pm[3];
#pragma unroll
        for (int s = 4; s < S; s += 4) {
            m0 = vmin(m0, pm[s]); m1 = vmin(m1, pm[s + 1]); m2 = vmin(m2, pm[s + 2]); m3 = vmin(m3, pm[s + 3]);
        }
        const double mn = vmin(vmin(m0, m1), vmin(m2, m3));
#ifdef ASMSCAN
        // four compares into four SGPR pairs, then four selects: no compare result is consumed by the next instruction
#pragma unroll
        for (int s = S - 4; s >= 0; s -= 4) {
            asm("v_cmp_eq_f64 %1, %5, %9\n\tv_cmp_eq_f64 %2, %6, %9\n\tv_cmp_eq_f64 %3, %7, %9\n\tv_cmp_eq_f64 %4, %8, %9\n\t"
                "v_cndmask_b32 %0, %0, %10, %1\n\tv_cndmask_b32 %0, %0, %11, %2\n\tv_cndmask_b32 %0, %0, %12, %3\n\t"
                "v_cndmask_b32 %0, %0, %13, %4"
                : "+v"(bst), "=&s"(k0), "=&s"(k1), "=&s"(k2), "=&s"(k3)
                : "v"(pm[rotl6(s + 3, R + 1)]), "v"(pm[rotl6(s + 2, R + 1)]), "v"(pm[rotl6(s + 1, R + 1)]), "v"(pm[rotl6(s, R + 1)]),
                  "v"(mn), "v"(s + 3), "v"(s + 2), "v"(s + 1), "v"(s));
        }
#else
#pragma unroll
        for (int s = S - 1; s >= 0; s--) bst = (pm[rotl6(s, R + 1)] == mn) ? s : bst;
#endif
    }
    *dec = ((unsigned long long)dw1 << 32) | dw0;
    *best = (unsigned char)bst;
}

template <bool ARGMIN, bool BM>
__global__ __launch_bounds__(64) void acs_cw_kernel(const double *__restrict__ llr, int64_t B, int64_t len, int64_t T,
                                                    unsigned long long *__restrict__ dec, unsigned char *__restrict__ best) {
    const int64_t cw = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (cw >= B) return;
    const double *x = llr + cw * len;
#ifdef ROWMAJOR
    unsigned long long *d = dec + cw * T;                                          // [B][T]: one row per codeword
    unsigned char *bs = best + cw * T;
#define STRIDE 1
#else
    unsigned long long *d = dec + (int64_t)blockIdx.x * T * 64 + threadIdx.x;     // [group][t][lane]
    unsigned char *bs = best + (int64_t)blockIdx.x * T * 64 + threadIdx.x;
#define STRIDE 64
#endif
    double pm[S];
#pragma unroll
    for (int s = 0; s < S; s++) pm[s] = (s == 0) ? 0.0 : INITV;
    double2 v[6];
#pragma unroll
    for (int u = 0; u < 6; u++) v[u] = *reinterpret_cast<const double2 *>(x + 2 * u);
    for (int64_t t = 0; t + 6 <= T; t += 6) {
        double2 nv[6];
#pragma unroll
        for (int u = 0; u < 6; u++) nv[u] = *reinterpret_cast<const double2 *>(x + 2 * (t + 6 + u));   // prefetch (row has slack)
        step<0, ARGMIN, BM>(pm, v[0].x, v[0].y, d + (t + 0) * STRIDE, bs + (t + 0) * STRIDE);
        step<1, ARGMIN, BM>(pm, v[1].x, v[1].y, d + (t + 1) * STRIDE, bs + (t + 1) * STRIDE);
        step<2, ARGMIN, BM>(pm, v[2].x, v[2].y, d + (t + 2) * STRIDE, bs + (t + 2) * STRIDE);
        step<3, ARGMIN, BM>(pm, v[3].x, v[3].y, d + (t + 3) * STRIDE, bs + (t + 3) * STRIDE);
        step<4, ARGMIN, BM>(pm, v[4].x, v[4].y, d + (t + 4) * STRIDE, bs + (t + 4) * STRIDE);
        step<5, ARGMIN, BM>(pm, v[5].x, v[5].y, d + (t + 5) * STRIDE, bs + (t + 5) * STRIDE);
#pragma unroll
        for (int u = 0; u < 6; u++) v[u] = nv[u];
    }
}

__global__ void fill_kernel(double *x, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)(i * 2654435761u) ^ (unsigned)(i >> 13);
        h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
        x[i] = ((double)(h & 0xffff) / 65536.0 - 0.5) * SCALE;
    }
}

template <bool ARGMIN, bool BM>
float run(const double *llr, int64_t B, int64_t len, int64_t T, unsigned long long *dec, unsigned char *best) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const unsigned nb = (unsigned)((B + 63) / 64);
    hipLaunchKernelGGL((acs_cw_kernel<ARGMIN, BM>), dim3(nb), dim3(64), 0, 0, llr, B, len, T, dec, best);
    CK(hipDeviceSynchronize());
    float best_ms = 1e9f;
    for (int it = 0; it < ITERS; it++) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((acs_cw_kernel<ARGMIN, BM>), dim3(nb), dim3(64), 0, 0, llr, B, len, T, dec, best);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best_ms) best_ms = ms;
    }
    return best_ms;
}

int main() {
    const int64_t len = LEN, T = TT;
    for (int64_t B : {65536ll, 131072ll}) {
        double *llr; unsigned long long *dec; unsigned char *best;
        CK(hipMalloc(&llr, sizeof(double) * B * len));
        CK(hipMalloc(&dec, 8 * B * T));
        CK(hipMalloc(&best, B * T));
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, llr, B * len);
        CK(hipDeviceSynchronize());
        const float full = run<true, true>(llr, B, len, T, dec, best);
        std::vector<unsigned char> hb((size_t)B * T);
        CK(hipMemcpy(hb.data(), best, hb.size(), hipMemcpyDeviceToHost));
        unsigned long long fnv = 1469598103934665603ull;
        for (unsigned char c : hb) { fnv ^= c; fnv *= 1099511628211ull; }
        const float noarg = run<false, true>(llr, B, len, T, dec, best);
        const float nobm = run<true, false>(llr, B, len, T, dec, best);
        const float core = run<false, false>(llr, B, len, T, dec, best);
        std::vector<unsigned long long> h(4);
        CK(hipMemcpy(h.data(), dec + 64 * 100, 32, hipMemcpyDeviceToHost));
        printf("B=%lld T=%lld: full %.3f ms | no argmin %.3f | no branch-metric math %.3f | ACS core only %.3f   (per step per wave, full: %.0f cycles @2.4GHz; waves/SIMD %.1f) chk %llx best-hash %llx\n",
               (long long)B, (long long)T, full, noarg, nobm, core, full * 1e-3 * 2.4e9 / T / ((B / 64 + 1023) / 1024), B / 64 / 1024.0, h[0] ^ h[3], fnv);
        CK(hipFree(llr)); CK(hipFree(dec)); CK(hipFree(best));
    }
    return 0;
}
