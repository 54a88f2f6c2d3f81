// General Viterbi decoder for gfx950: every trellis the reference's Trellis class can describe and every traceback
// depth -- the slow-but-complete path behind the specialised kernels of viterbi.hip / viterbi_cw.hip.
// Replaces, for the argument domain those kernels refuse (more than 128 states, k > 2, n > 6, traceback windows whose
// ring does not fit the 64 KiB of a wavefront's LDS), the body of
//   viterbi_decode / _acs_traceback / _where_c / _compute_branch_metrics
//   (/root/reference/commpy/channelcoding/convcode.py:661-749, :590-657, :561-572, :575-587)
// with the same decision rule as every other kernel of the library:
//   forward add-compare-select over t = 1..T, float64 path metrics, np.where order of the predecessors, NumPy's
//   min / argmin semantics (first minimum; a NaN candidate wins, :633-637, :645);
//   best[t] = argmin state;  symbol of step s = survivor symbol at step s of the path traced back from
//   best[min(s + tb - 2, T)]  (every traceback of :644-654 rewrites steps t-tb+2 .. t, the last writer is that one).
//
// Mapping: ONE WORKGROUP PER CODEWORD (64 threads up to 64 states, else 256), persistent over the batch.
//   * thread = state (strided when S > workgroup size); path metrics double-buffered in LDS up to 2048 states, in an
//     L2-resident HBM scratch above;
//   * the n per-bit metrics of a step are evaluated once by n threads (the reference's two operations per bit) and every
//     branch adds them in NumPy's add.reduce order (n < 8 sequential, else eight accumulators);
//   * survivor decisions leave as k ballot words per 64 states (bit planes of the chosen predecessor index) into a ring
//     [slot][k][S/64] in HBM scratch, best[t] into a ring [slot];
//   * every NT steps (and at the end) each thread owns one output step and walks its tb - 2 hops through the ring.
// Speed is not the goal (a K = 9 code decodes ~0.3 M info-bit/s per workgroup, the reference 1 k); HBM traffic beyond
// the algorithmic bytes stays in L2.
#include "cpx_internal.h"
#include "cpx_math.h"

#include <algorithm>

using namespace cpx;

namespace {

struct GenParams {
    const double *coded;       // [B][len]
    uint8_t *bits;             // [B][L]
    const int32_t *pred_state, *pred_input, *pred_code;   // [S][I], np.where order
    int64_t B, len, L, T, Lk;
    int k, n, S, I, type, tb, RS, W;
    unsigned long long *dring; // per workgroup [RS][k][W]: bit plane b of the decisions of 64 consecutive states
    int32_t *bring;            // per workgroup [RS]: best state of the step
    double *pm;                // per workgroup [2][S], or null: the metrics live in LDS
};

// per-bit metrics of one received value (convcode.py:575-587): m0 = cost of code bit 0, m1 = of bit 1
__device__ __forceinline__ void gen_bit_metrics(int type, double r, double &m0, double &m1) {
    if (type == CPX_VIT_HARD) {
        const long long ri = (long long)r;            // r_codeword.astype(int) (:580)
        m0 = (double)(ri ^ 0ll);                      // hamming_dist = sum of xor (utilities.py:130)
        m1 = (double)(ri ^ 1ll);
    } else if (type == CPX_VIT_SOFT) {
        const double nll0 = fast_log<true>(exp(r) + 1.0);   // :582
        m0 = nll0;
        m1 = nll0 - r;                                // :583
    } else {
        const double d0 = r - (-1.0), d1 = r - 1.0;   // i_codeword_array = 2*c - 1 (:586), euclid_dist utilities.py:152
        m0 = d0 * d0;
        m1 = d1 * d1;
    }
}

// sum over the n bits of codeword `code` (MSB = output 0, :622) in NumPy's float64 add.reduce order (:584, utilities.py:152)
__device__ __forceinline__ double branch_metric(const double *bm, int code, int n) {
    auto a = [&](int j) { return bm[(((code >> (n - 1 - j)) & 1) << 4) + j]; };
    if (n < 8) {
        double res = 0.0;
        for (int j = 0; j < n; j++) res += a(j);
        return res;
    }
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = a(j);
    int i = 8;
    for (; i < n - (n % 8); i += 8)
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] += a(i + j);
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a(i);
    return res;
}

// NumPy's argmin order over (value, index): a NaN beats every number, among equals the lower index wins
__device__ __forceinline__ bool np_before(double av, int ai, double bv, int bi) {
    const bool an = av != av, bn = bv != bv;
    if (an != bn) return an;
    if (an) return ai < bi;
    return av < bv || (av == bv && ai < bi);
}

template <int NT>
__global__ __launch_bounds__(NT) void viterbi_generic_kernel(GenParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double *bm = reinterpret_cast<double *>(smem_raw);           // [2][16] per-bit metrics of the step
    double *redv = bm + 32;                                       // [4] wave minima
    int *redi = reinterpret_cast<int *>(redv + 4);                // [4] their states
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int S = p.S, I = p.I, k = p.k, n = p.n, W = p.W, RM = p.RS - 1;
    double *pm0, *pm1;
    if (p.pm) { pm0 = p.pm + (size_t)blockIdx.x * 2 * S; pm1 = pm0 + S; }
    else { pm0 = redv + 8; pm1 = pm0 + S; }
    unsigned long long *dring = p.dring + (size_t)blockIdx.x * p.RS * k * W;
    int32_t *bring = p.bring + (size_t)blockIdx.x * p.RS;
    const double pad = (p.type == CPX_VIT_UNQUANTIZED) ? -1.0 : 0.0;  // :726-732
    const int Sr = (S + NT - 1) / NT * NT;

    for (int64_t cw = blockIdx.x; cw < p.B; cw += gridDim.x) {
        for (int s = tid; s < S; s += NT) pm0[s] = (s == 0) ? 0.0 : __builtin_huge_val();   // :705-706
        __syncthreads();
        const double *rx = p.coded + cw * p.len;
        int64_t next_out = 1, chunk0 = 1;
        for (int64_t t = 1; t <= p.T; t++) {
            if (tid < n) {
                double r = (t <= p.Lk) ? rx[(t - 1) * n + tid] : pad;                 // :723-732
                if (p.type == CPX_VIT_SOFT) r = r < -500.0 ? -500.0 : (r > 500.0 ? 500.0 : r);   // :719 (a NaN stays)
                double m0, m1;
                gen_bit_metrics(p.type, r, m0, m1);
                bm[tid] = m0;
                bm[16 + tid] = m1;
            }
            __syncthreads();
            const int slot = (int)(t & RM);
            double bv = __builtin_huge_val();
            int bi = 0x7fffffff;
            for (int s0 = 0; s0 < Sr; s0 += NT) {
                const int s = s0 + tid;
                const bool valid = s < S;
                double mv = 0.0;
                int mi = 0;
                if (valid) {
                    const int32_t *ps = p.pred_state + (size_t)s * I, *pc = p.pred_code + (size_t)s * I;
                    mv = pm0[ps[0]] + branch_metric(bm, pc[0], n);                     // :629
                    for (int i = 1; i < I; i++) {
                        const double c = pm0[ps[i]] + branch_metric(bm, pc[i], n);
                        // pmetrics.min() / .argmin() (:633-637): first minimum, a NaN wins and stays
                        if (!(mv != mv) && (c < mv || c != c)) { mv = c; mi = i; }
                    }
                    pm1[s] = mv;
                    if (np_before(mv, s, bv, bi)) { bv = mv; bi = s; }
                }
                for (int b = 0; b < k; b++) {
                    const unsigned long long w = __ballot(valid && ((mi >> b) & 1));
                    if (lane == 0 && s < S) dring[((size_t)slot * k + b) * W + (s >> 6)] = w;
                }
            }
            // current_state = path_metrics[:, 1].argmin() (:645)
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const double ov = __shfl_xor(bv, d);
                const int oi = __shfl_xor(bi, d);
                if (np_before(ov, oi, bv, bi)) { bv = ov; bi = oi; }
            }
            if (lane == 0) { redv[wave] = bv; redi[wave] = bi; }
            __syncthreads();
            if (tid == 0) {
                double v = redv[0];
                int ix = redi[0];
                for (int w2 = 1; w2 < NT / 64; w2++)
                    if (np_before(redv[w2], redi[w2], v, ix)) { v = redv[w2]; ix = redi[w2]; }
                bring[slot] = ix;
            }
            { double *tp = pm0; pm0 = pm1; pm1 = tp; }                                  // :746-747
            __syncthreads();
            if (t - chunk0 + 1 == NT || t == p.T) {
                chunk0 = t + 1;
                const int64_t s_hi = (t >= p.T) ? p.T : (t - p.tb + 2);
                auto decision = [&](int64_t tt, int st) {
                    const size_t base = (size_t)(tt & RM) * k * W + (st >> 6);
                    int j = 0;
                    for (int b = 0; b < k; b++) j |= (int)((dring[base + (size_t)b * W] >> (st & 63)) & 1ull) << b;
                    return j;
                };
                while (next_out <= s_hi) {
                    const int64_t so = next_out + tid;
                    if (so <= s_hi) {
                        int64_t t0 = so + p.tb - 2;
                        if (t0 > p.T) t0 = p.T;
                        int st = bring[(int)(t0 & RM)];
                        for (int64_t tt = t0; tt > so; --tt) st = p.pred_state[(size_t)st * I + decision(tt, st)];   // :650
                        const int sym = p.pred_input[(size_t)st * I + decision(so, st)];                             // :649
                        for (int b = 0; b < k; b++) {                                                               // :651-653
                            const int64_t pos = (so - 1) * k + b;
                            if (pos < p.L) p.bits[cw * p.L + pos] = (uint8_t)((sym >> (k - 1 - b)) & 1);
                        }
                    }
                    next_out = (next_out + NT <= s_hi + 1) ? next_out + NT : s_hi + 1;
                }
                __syncthreads();
            }
        }
    }
}

int next_pow2_64(int64_t v) {
    int64_t r = 1;
    while (r < v) r <<= 1;
    return (int)r;
}

}  // namespace

namespace cpx {

int viterbi_generic(const cpx_trellis *t, const double *d_coded, int64_t B, int64_t len, int64_t L, int64_t T, int tb, int type,
                    uint8_t *d_bits, hipStream_t st) {
    CPX_REQUIRE(t->k <= 8, CPX_ELIMIT, "viterbi: k = %d > 8 not supported", t->k);
    CPX_REQUIRE(t->n <= 16, CPX_ELIMIT, "viterbi: n = %d > 16 not supported", t->n);
    GenParams p;
    p.coded = d_coded; p.bits = d_bits;
    p.pred_state = t->d_pred_state; p.pred_input = t->d_pred_input; p.pred_code = t->d_pred_code;
    p.B = B; p.len = len; p.L = L; p.T = T; p.Lk = L / t->k;
    p.k = t->k; p.n = t->n; p.S = t->S; p.I = t->I; p.type = type; p.tb = tb;
    const int NT = (t->S <= 64) ? 64 : 256;
    p.W = (t->S + 63) / 64;
    const int64_t window = std::min<int64_t>((int64_t)NT + tb, T + 1);   // steps a traceback chunk can reach back to
    CPX_REQUIRE(window < (1ll << 24), CPX_ELIMIT, "viterbi: traceback window of %lld steps not supported", (long long)window);
    p.RS = next_pow2_64(window);
    const bool lds_pm = t->S <= 2048;                               // 32 KiB of metrics: no opt-in for dynamic LDS above 64 KiB needed
    const size_t lds = 8 * (32 + 8) + (lds_pm ? sizeof(double) * 2 * (size_t)t->S : 0);
    // persistent grid: what fits the scratch budget (256 MB of decision ring), at most two workgroups per SIMD's worth
    const size_t ring_bytes = sizeof(unsigned long long) * (size_t)p.RS * p.k * p.W + sizeof(int32_t) * (size_t)p.RS;
    const size_t pm_bytes = lds_pm ? 0 : sizeof(double) * 2 * (size_t)t->S;
    int64_t grid = std::min<int64_t>(B, (int64_t)device_cus() * (NT == 64 ? 16 : 4));
    grid = std::min<int64_t>(grid, std::max<int64_t>(1, (int64_t)(((size_t)1 << 28) / (ring_bytes + pm_bytes))));
    // scratch-arena slots 8 .. 10 (6 / 7 are the host-buffer pipeline's staging blocks, viterbi.hip)
    void *w0 = nullptr, *w1 = nullptr, *w2 = nullptr;
    if (int rc = workspace(st, 8, sizeof(unsigned long long) * (size_t)p.RS * p.k * p.W * (size_t)grid, &w0)) return rc;
    if (int rc = workspace(st, 9, sizeof(int32_t) * (size_t)p.RS * (size_t)grid, &w1)) return rc;
    if (!lds_pm) if (int rc = workspace(st, 10, pm_bytes * (size_t)grid, &w2)) return rc;
    p.dring = static_cast<unsigned long long *>(w0);
    p.bring = static_cast<int32_t *>(w1);
    p.pm = static_cast<double *>(w2);
    if (NT == 64) hipLaunchKernelGGL((viterbi_generic_kernel<64>), dim3((unsigned)grid), dim3(64), lds, st, p);
    else hipLaunchKernelGGL((viterbi_generic_kernel<256>), dim3((unsigned)grid), dim3(256), lds, st, p);
    CPX_HIP(hipGetLastError());
    note_kernel("viterbi_generic_kernel<%d> (%d states, k = %d, n = %d, ring %d)", NT, t->S, t->k, t->n, p.RS);
    return CPX_OK;
}

}  // namespace cpx
