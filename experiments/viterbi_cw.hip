// EXPERIMENT -- NOT BUILT INTO libcommpy_amd.so (see DESIGN.md 4.1 "tried and rejected").
// Measured on MI355X, BASELINE config 2 (B = 65536, K = 7 soft): ACS kernel 3.14 ms + traceback kernel 0.56 ms = 3.7 ms
// against 3.2 ms for the state-per-lane kernel of commpy_amd/csrc/viterbi.hip.  Bit-exact against the oracle in
// every test, but (a) the straight-line ACS of 64 states runs at one wavefront per SIMD (256 VGPR + 38 AGPR) and
// its dependent float64 add -> compare -> min chains stall without a second wavefront to fill the gaps
// (7.3k cycles per trellis step for 64 codewords against a 3.0k-cycle issue model), and (b) one build of the
// traceback kernel returned wrong bits for workgroups >= 1024 of a single launch although its inputs in HBM were
// verified correct (a host emulation of the same walk on the dumped arrays matched the reference); launching it
// in chunks of 1024 workgroups, or an equivalent source change that altered the scalar register allocation, made
// the problem disappear.  Not understood => not shipped.
//
// Viterbi decoder, "codeword per lane" variant for large batches of the standard rate-1/2 feed-forward codes
// (same contract and decision rule as viterbi.hip: convcode.py:661-749, :590-657, :575-587).
//
// viterbi.hip maps one trellis STATE to a lane: every step pays a cross-lane exchange of the path metrics, an
// all-lane float64 minimum (first-argmin rule) and a ballot -- 89 issue cycles per trellis step per codeword on
// MI355X, VALU-bound.  When the batch is large enough to give every SIMD a wavefront of 64 codewords
// (B >= 32768), a lane can own a whole CODEWORD instead:
//   * the S path metrics of the codeword live in the lane's registers (2 x S float64 with double buffering);
//     a shift-register trellis has static predecessors (2(s mod S/2), +1), so the add-compare-select of all S
//     states is straight-line code with NO cross-lane traffic: 2 v_add_f64 + v_min_f64 + v_cmp/v_addc per state;
//   * WHICH of the four branch metrics a branch uses depends on the generator polynomials; they are template
//     parameters (the branch code is a constexpr function), instantiated for the standard codes below and checked
//     against the trellis tables the caller built -- any other trellis takes the state-per-lane kernels;
//   * the first-argmin state of a step is an in-lane minimum + an in-lane first-equal scan;
//   * decisions (one bit per state) and the first-argmin state of every step go to a workspace in HBM
//     (9 B per trellis step), and a second kernel runs the sliding traceback lane-parallel over output steps,
//     exactly like viterbi.hip does from its LDS ring.
// Same arithmetic in the same order as viterbi.hip (float64, -ffp-contract=off): bit-identical output.
#include "cpx_internal.h"

#include <algorithm>
#include <cstdio>
#include <vector>

using namespace cpx;

namespace {

template <int LGS, unsigned G0, unsigned G1>
struct SrCode {
    static constexpr int S = 1 << LGS;
    static constexpr int parity(unsigned v) { return __builtin_popcount(v) & 1; }
    // 2-bit output (MSB = first generator) of the branch into state s from its j-th predecessor:
    // register = [input bit | predecessor state], convcode.py:166-175 (generator MSB taps the input)
    static constexpr int code(int s, int j) {
        const unsigned p = (unsigned)(((s << 1) & (S - 1)) | j), b = (unsigned)(s >> (LGS - 1));
        const unsigned reg = (b << LGS) | p;
        return (parity(reg & G0) << 1) | parity(reg & G1);
    }
};

__device__ __forceinline__ double vmin(double a, double b) {      // one v_min_f64 (fmin adds canonicalising v_max)
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// acc = 2*acc + (x < y): the decision bit of a state shifted into a 32-bit word (v_cmp + v_addc)
__device__ __forceinline__ unsigned shift_in_lt(unsigned acc, double x, double y) {
    unsigned r;
    asm("v_cmp_lt_f64 vcc, %2, %3\n\tv_addc_co_u32 %0, vcc, %1, %1, vcc" : "=v"(r) : "v"(acc), "v"(x), "v"(y) : "vcc");
    return r;
}

struct CwParams {
    const double *coded;          // [B][len]
    unsigned long long *dec;      // [B][T]  decision word of step t (bit layout: see dec_bit)
    unsigned char *best;          // [B][T]  first-argmin state of step t
    uint8_t *bits;                // [B][L]
    int64_t B, len, L, T, Lk;
    int type, tb, RSW;
};

// decision words: states 32w .. 32w+31 are shifted into 32-bit word w in increasing order, so state s sits at bit
// (min(S,32) - 1 - (s & 31)) of word s >> 5
template <int LGS>
__device__ __forceinline__ int dec_bit(unsigned long long w, int st) {
    constexpr int S = 1 << LGS, WB = S < 32 ? S : 32;
    return (int)((w >> (((st >> 5) << 5) + (WB - 1 - (st & 31)))) & 1ull);
}

// Per-bit metrics of one received value (convcode.py:575-587) -- identical to viterbi.hip
__device__ __forceinline__ void bit_metrics_cw(int type, double r, double &m0, double &m1) {
    if (type == CPX_VIT_HARD) {
        long long ri = (long long)r;
        m0 = (double)(ri ^ 0ll);
        m1 = (double)(ri ^ 1ll);
    } else if (type == CPX_VIT_SOFT) {
        double nll0 = log(exp(r) + 1.0);
        m0 = nll0;
        m1 = nll0 - r;
    } else {
        double d0 = r - (-1.0), d1 = r - 1.0;
        m0 = d0 * d0;
        m1 = d1 * d1;
    }
}

template <int LGS, unsigned G0, unsigned G1>
__global__ __launch_bounds__(64) void viterbi_cw_acs_kernel(CwParams p) {
    using C = SrCode<LGS, G0, G1>;
    constexpr int S = 1 << LGS, W = S > 32 ? S / 32 : 1;
    const int64_t cw = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool valid = cw < p.B;
    const double *x = p.coded + (valid ? cw : 0) * p.len;
    unsigned long long *dec = p.dec + (valid ? cw : 0) * p.T;
    unsigned char *best = p.best + (valid ? cw : 0) * p.T;

    double pa[S], pb[S];
#pragma unroll
    for (int s = 0; s < S; s++) pa[s] = (s == 0) ? 0.0 : __builtin_huge_val();   // path_metrics[:,0] = inf, [0][0] = 0 (:705-706)

    auto load = [&](int64_t t, double &r0, double &r1) {
        const bool have = valid && t <= p.Lk && t <= p.T;             // t > L//k -> padding (:722-734)
        r0 = r1 = (p.type == CPX_VIT_UNQUANTIZED) ? -1.0 : 0.0;
        if (have) {
            const double2 v = *reinterpret_cast<const double2 *>(x + (t - 1) * 2);
            r0 = v.x; r1 = v.y;
        }
    };
    // one trellis step: src -> dst, decision word and first-argmin state stored for step t
    auto step = [&](const double (&src)[S], double (&dst)[S], int64_t t, double r0, double r1) {
        if (p.type == CPX_VIT_SOFT) {                                  // coded_bits.clip(-500, 500) (:719)
            r0 = fmin(fmax(r0, -500.0), 500.0);
            r1 = fmin(fmax(r1, -500.0), 500.0);
        }
        double m00, m01, m10, m11;
        bit_metrics_cw(p.type, r0, m00, m01);
        bit_metrics_cw(p.type, r1, m10, m11);
        double bmv[4];                                                 // NumPy add.reduce, n < 8: sequential from 0
        bmv[0] = (0.0 + m00) + m10; bmv[1] = (0.0 + m00) + m11;
        bmv[2] = (0.0 + m01) + m10; bmv[3] = (0.0 + m01) + m11;
        unsigned dw[W];
#pragma unroll
        for (int w = 0; w < W; w++) dw[w] = 0;
#pragma unroll
        for (int s = 0; s < S; s++) {
            const int p0 = (s << 1) & (S - 1);
            const double a0 = src[p0] + bmv[C::code(s, 0)];            // pmetrics[0] (:629)
            const double a1 = src[p0 | 1] + bmv[C::code(s, 1)];
            dw[s >> 5] = shift_in_lt(dw[s >> 5], a1, a0);              // first minimum wins (:633-642)
            dst[s] = vmin(a0, a1);
        }
        // first-argmin state (:645)
        double m0 = dst[0], m1 = dst[1 % S], m2 = dst[2 % S], m3 = dst[3 % S];
#pragma unroll
        for (int s = 4; s < S; s += 4) {
            m0 = vmin(m0, dst[s]); m1 = vmin(m1, dst[s + 1]); m2 = vmin(m2, dst[s + 2]); m3 = vmin(m3, dst[s + 3]);
        }
        const double mn = vmin(vmin(m0, m1), vmin(m2, m3));
        int bst = 0;
#pragma unroll
        for (int s = S - 1; s >= 0; s--) bst = (dst[s] == mn) ? s : bst;
        if (valid) {
            unsigned long long word = dw[0];
            if (W == 2) word |= (unsigned long long)dw[W - 1] << 32;
            dec[t - 1] = word;
            best[t - 1] = (unsigned char)bst;
        }
    };

    double r0, r1, n0, n1;
    load(1, r0, r1);
    int64_t t = 1;
    for (; t + 1 <= p.T; t += 2) {
        load(t + 1, n0, n1);
        step(pa, pb, t, r0, r1);
        load(t + 2, r0, r1);
        step(pb, pa, t + 1, n0, n1);
    }
    if (t <= p.T) step(pa, pb, t, r0, r1);
}

// Sliding traceback, one wavefront per codeword, lane-parallel over output steps (lane i owns output step
// base + i): bit of step s = survivor symbol at step s of the path traced back from best[min(s + tb - 2, T)].
template <int LGS>
__global__ __launch_bounds__(64) void viterbi_cw_tb_kernel(CwParams p) {
    constexpr int S = 1 << LGS;
    __shared__ unsigned long long win[512];                                       // [RSW <= 512] ring of decision words
    const int lane = threadIdx.x, RM = p.RSW - 1;
    const int64_t cw = (int64_t)blockIdx.x + p.Lk;                                 // DEBUG: Lk reused as block offset in TB
    const unsigned long long *dec = p.dec + cw * p.T;
    const unsigned char *best = p.best + cw * p.T;
    int64_t loaded = 0;                                                           // steps 1..loaded are in the ring
    for (int64_t base = 1; base <= p.T; base += 64) {
        int64_t need = base + 63 + p.tb - 2;
        if (need > p.T) need = p.T;
        __syncthreads();                                                          // readers of the previous window are done
        for (int64_t tt = loaded + 1 + lane; tt <= need; tt += 64)
            win[(int)(tt & RM)] = dec[tt - 1];
        loaded = need;
        __syncthreads();
        const int64_t so = base + lane;
        if (so <= p.T) {
            int64_t t0 = so + p.tb - 2;
            if (t0 > p.T) t0 = p.T;
            int st = (int)best[t0 - 1];
            for (int64_t tt = t0; tt > so; --tt) st = ((st << 1) & (S - 1)) | dec_bit<LGS>(win[(int)(tt & RM)], st);
            const int64_t pos = so - 1;
            if (pos < p.L) p.bits[cw * p.L + pos] = (uint8_t)(st >> (LGS - 1));    // input bit of the branch into st
        }
    }
}

template <int LGS, unsigned G0, unsigned G1>
bool tables_match(const cpx_trellis *t) {
    using C = SrCode<LGS, G0, G1>;
    if (t->S != (1 << LGS) || t->I != 2 || t->k != 1 || t->n != 2) return false;
    for (int s = 0; s < t->S; s++)
        for (int j = 0; j < 2; j++) {
            if (t->pred_state[s * 2 + j] != (((s << 1) & (t->S - 1)) | j)) return false;
            if (t->pred_input[s * 2 + j] != (s >> (LGS - 1))) return false;
            if (t->pred_code[s * 2 + j] != C::code(s, j)) return false;
        }
    return true;
}

template <int LGS, unsigned G0, unsigned G1>
void launch(const CwParams &p, hipStream_t st) {
    const unsigned nb = (unsigned)((p.B + 63) / 64);
    hipLaunchKernelGGL((viterbi_cw_acs_kernel<LGS, G0, G1>), dim3(nb), dim3(64), 0, st, p);
    { CwParams q = p; q.Lk = 0; hipLaunchKernelGGL((viterbi_cw_tb_kernel<LGS>), dim3((unsigned)p.B), dim3(64), 0, st, q); }
}

}  // namespace

namespace cpx {

// Returns true when the call was handled here (rc = status); false -> the caller uses the state-per-lane kernels.
bool viterbi_codeword_path(const cpx_trellis *t, const double *d_coded, int64_t B, int64_t len, int64_t L, int64_t T,
                           int tb, int type, uint8_t *d_bits, hipStream_t st, int *rc) {
    *rc = CPX_OK;
    const char *e = getenv("CPX_VITERBI_PATH");                   // "cw" / "cw!" / "wave" force a path (tests, experiments)
    if (e && e[0] == 'w') return false;
    const bool forced = e && e[0] == 'c';
    if (!forced && B < 32768) return false;                        // fewer than half a wavefront per SIMD: the wave kernels win
    if (t->I != 2 || t->k != 1 || t->n != 2 || B >= (1ll << 31) || T < 1) return false;
    int rsw = 64;
    while (rsw < 64 + tb) rsw <<= 1;
    if (rsw > 512) return false;
    CwParams p;
    p.coded = d_coded; p.bits = d_bits; p.B = B; p.len = len; p.L = L; p.T = T; p.Lk = L;   // k = 1
    p.type = type; p.tb = tb; p.RSW = rsw;
#define CPX_TRY(LG, GA, GB)                                                                                        \
    if (tables_match<LG, GA, GB>(t)) {                                                                             \
        if ((*rc = workspace(st, 0, sizeof(unsigned long long) * (size_t)(B * T), (void **)&p.dec))) return true;  \
        if ((*rc = workspace(st, 1, (size_t)(B * T), (void **)&p.best))) return true;                              \
        launch<LG, GA, GB>(p, st);                                                                                 \
        if (hipGetLastError() != hipSuccess) { set_error("viterbi (codeword path): launch failed"); *rc = CPX_EHIP; } \
        return true;                                                                                               \
    }
    // Template generators are in "MSB taps the input" order.  commpy's default polynomial_format='MSB' makes the
    // LEAST significant bit of the octal number the D^0 tap (convcode.py:211-222), i.e. the bit-reversed number here:
    // (133,171) -> (155,117), (23,35) -> (31,27); polynomial_format='Matlab' keeps the number as written.
    CPX_TRY(6, 0155u, 0117u)      // K = 7 (133,171), commpy default format: 802.11 / BASELINE configs 2 and 5
    CPX_TRY(6, 0117u, 0155u)      // K = 7 (171,133)
    CPX_TRY(6, 0133u, 0171u)      // K = 7 (133,171), Matlab format
    CPX_TRY(2, 05u, 07u)          // K = 3 (5,7): BASELINE config 1, the reference's own test code (palindromes)
    CPX_TRY(2, 07u, 05u)
    CPX_TRY(4, 031u, 027u)        // K = 5 (23,35), commpy default format
    CPX_TRY(4, 023u, 035u)        // K = 5 (23,35), Matlab format
#undef CPX_TRY
    if (forced && e[1] == 'w' && e[2] == '!') {                   // "cw!": tests insist on this path
        set_error("viterbi (codeword path): no instantiation for this trellis");
        *rc = CPX_ELIMIT;
        return true;
    }
    return false;
}

}  // namespace cpx
