// Viterbi decoder for gfx950 -- one wavefront per codeword (S = 64), or 64/S codewords per
// wavefront for smaller trellises.  Replaces the body of the reference's
//   viterbi_decode / _acs_traceback / _compute_branch_metrics
//   (/root/reference/commpy/channelcoding/convcode.py:661-749, :590-657, :575-587)
// with the decision rule of SURVEY Appendix A.1 (verified against the live reference):
//   forward ACS over t = 1..T with float64 path metrics, first-minimum tie rule in np.where order;
//   best[t] = first-argmin state;  bit(s) of step s = survivor symbol at step s of the path traced
//   back from best[min(s + tb - 2, T)].
//
// Mapping (wave64):
//   lane = g*S + s : codeword slot g (G = 64/S slots per wave), trellis state s.
//   * path metric of state s lives in one VGPR pair of lane s; the predecessors' metrics are
//     exchanged across lanes through a 512-byte LDS buffer (ds_write_b64 + one ds_read_b128 per
//     step; measured 2.4x cheaper on the LDS pipe than the four ds_bpermute_b32 of a float64
//     shuffle pair, which bounded the first versions of this kernel);
//   * branch metrics: per chunk of CH = S steps, lane (g,i) loads the n received values of step
//     t_base+i (coalesced, prefetched one chunk ahead), evaluates the reference's per-bit metrics
//     once and writes the 2^n codeword metrics of that step to LDS; the ACS lanes read them back
//     by codeword index (wave-uniform broadcasts);
//   * first-argmin state per step: all-lane minimum by DPP row permutes + v_permlane16/32_swap
//     (no LDS); for S = 64 eight consecutive steps share ONE "transposed" reduction tree;
//   * survivor decisions: one 64-bit ballot word per step (two for I = 4); lane i keeps the word
//     and the argmin of step t_base+i in registers and the chunk is flushed to an LDS ring of
//     RS >= CH + tb - 2 steps with one store per lane;
//   * sliding traceback after every chunk, lane-parallel over output steps (lane (g,i) owns
//     output step next_out + i of codeword g), walking tb-2 decision words of the ring.
// HBM traffic is exactly the algorithmic one: each received value is read once, each decoded
// bit written once.  float64 throughout (the reference's arithmetic); compiled with
// -ffp-contract=off so sums round like NumPy's.
#include "cpx_internal.h"
#include "cpx_math.h"
#include "demod_dev.h"

#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

using namespace cpx;

namespace {

struct VitParams {
    const double *coded;   // [B][len]
    uint8_t *bits;         // [B][L]
    const int32_t *pred_state, *pred_input, *pred_code;  // [S][I]
    int64_t B, len, L, T, Lk;
    int k, n, lgS, I, NC, type, tb, RS;
    // fused hard demodulation (cpx_demod_hard_viterbi_batch_dev): the kernel reads received SYMBOLS and takes the
    // hard decisions itself -- the int8 bits of Modem.demodulate(y, 'hard') never exist in HBM
    const double2 *ysym;   // [B][nsym] complex128, or null = `coded` holds the decoder input
    const double2 *cst;    // [M] constellation
    const double *axes;    // separable constellations: [2][sqrt(M)], else null
    int64_t nsym;
    int M, nb, nh;         // nh: log2(sqrt(M)) for separable constellations, 0 = generic scan
    // 'soft' only: one byte per wavefront-sized work item (`blk` of wave_body / the codeword of viterbi_wide_kernel), set
    // when one of its received values is NaN -- see "NaN among 'soft' inputs" below; null otherwise
    uint8_t *nanflags;
};

// ---- NaN among 'soft' inputs: detect and redo ---------------------------------------------------------------------
// The reference's clip lets a NaN through (convcode.py:719): every branch metric of that step is NaN, so is every path
// metric from then on, every comparison is false and `argmin` of all-NaN candidates is 0 (:633-645) -- from the first NaN
// step of a codeword on, ALL decisions are "first predecessor" and EVERY traceback starts from state 0.  Carrying that
// rule in the hot loops cost the fused kernel 2 % and the state-per-lane kernels 4 - 7 % (round 2) for an input no
// demodulator produces, so the fast kernels only DETECT it -- one unordered compare per received pair, OR-ed into a
// per-item flag byte; their arithmetic runs on with the NaN clipped to -500 -- and a second launch re-decodes the flagged
// items with the NaN-exact instantiation (NANX) of the state-per-lane body.  With no NaN in the batch that launch is
// ceil(items / 64) wavefronts that read 64 flag bytes each and exit.

// ---- cross-lane helpers -------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// min of two non-NaN doubles as ONE v_min_f64 (the fmin builtin adds two canonicalising v_max_f64).
// The trailing s_nop covers the VALU-write -> DPP / v_permlane read hazard of the consumer, which
// hipcc does not pad for an asm statement.
__device__ __forceinline__ double min_f64(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2\n\ts_nop 1" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ double mk(unsigned hi, unsigned lo) { return __hiloint2double((int)hi, (int)lo); }

// v_permlane16_swap(a, b): a' = [a.row0, b.row0, a.row2, b.row2], b' = [a.row1, b.row1, a.row3, b.row3]
// v_permlane32_swap(a, b): a' = [a.low half, b.low half],          b' = [a.high half, b.high half]
// With a == b == v the element-wise min of the two results is the all-reduce over the row pair / halves.
__device__ __forceinline__ double rowpair_min(double v) {
    const unsigned lo = __double2loint(v), hi = __double2hiint(v);
    const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return min_f64(mk(h[0], l[0]), mk(h[1], l[1]));
}

__device__ __forceinline__ double halves_min(double v) {
    const unsigned lo = __double2loint(v), hi = __double2hiint(v);
    const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return min_f64(mk(h[0], l[0]), mk(h[1], l[1]));
}

// All-reduce minimum inside each aligned group of 2^LGS lanes: DPP quad/row permutes for the four
// in-row stages, v_permlane16_swap / v_permlane32_swap (gfx950) across rows.  No LDS traffic.
template <int LGS>
__device__ __forceinline__ double group_min(double v) {
    if (LGS >= 1) v = min_f64(v, dpp_f64<0xB1>(v));    // quad_perm [1,0,3,2]  (lane ^ 1)
    if (LGS >= 2) v = min_f64(v, dpp_f64<0x4E>(v));    // quad_perm [2,3,0,1]  (lane ^ 2)
    if (LGS >= 3) v = min_f64(v, dpp_f64<0x141>(v));   // row_half_mirror      (other quad of the 8)
    if (LGS >= 4) v = min_f64(v, dpp_f64<0x140>(v));   // row_mirror           (other half of the row)
    if (LGS >= 5) v = rowpair_min(v);
    if (LGS >= 6) v = halves_min(v);
    return v;
}

// ---- eight all-lane minima for the price of ~1.5 (S = 64) ------------------------------------------
// Reduce EIGHT 64-lane vectors at once ("transposed" tree): every stage halves the number of live
// vectors instead of repeating the full tree per vector.
//   pair_halves(a, b):  v_permlane32_swap -> lanes 0-31: min(a[l], a[l+32]);  lanes 32-63: same for b
//   pair_rows(a, b):    v_permlane16_swap -> rows 0,2 carry a reduced over row pairs, rows 1,3 carry b
//   pair_octets(a, b):  two bank-masked DPP row shifts -> lanes 0-7 of each row carry a, 8-15 carry b
// After the three pairing stages ONE vector holds all eight partial minima (8 lanes each); three
// plain DPP stages finish them.  The minimum of input vector u ends up in lanes 8*bitrev3(u)..+7.
__device__ __forceinline__ double pair_halves(double a, double b) {
    const auto l = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto h = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return min_f64(mk(h[0], l[0]), mk(h[1], l[1]));
}

__device__ __forceinline__ double pair_rows(double a, double b) {
    const auto l = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto h = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return min_f64(mk(h[0], l[0]), mk(h[1], l[1]));
}

__device__ __forceinline__ double pair_octets(double a, double b) {
    // P = [a.lanes0-7 | b.lanes0-7 moved up by 8] (row_shr:8 into banks 2,3);  Q = [a.lanes8-15 moved down | b.lanes8-15]
    const int plo = __builtin_amdgcn_update_dpp(__double2loint(a), __double2loint(b), 0x118, 0xf, 0xc, false);
    const int phi = __builtin_amdgcn_update_dpp(__double2hiint(a), __double2hiint(b), 0x118, 0xf, 0xc, false);
    const int qlo = __builtin_amdgcn_update_dpp(__double2loint(b), __double2loint(a), 0x108, 0xf, 0x3, false);
    const int qhi = __builtin_amdgcn_update_dpp(__double2hiint(b), __double2hiint(a), 0x108, 0xf, 0x3, false);
    return min_f64(__hiloint2double(phi, plo), __hiloint2double(qhi, qlo));
}

__device__ __forceinline__ double min8_transposed(const double (&h)[8]) {
    const double c0 = pair_halves(h[0], h[1]), c1 = pair_halves(h[2], h[3]);
    const double c2 = pair_halves(h[4], h[5]), c3 = pair_halves(h[6], h[7]);
    const double d0 = pair_rows(c0, c1), d1 = pair_rows(c2, c3);
    double e = pair_octets(d0, d1);
    e = min_f64(e, dpp_f64<0x141>(e));     // row_half_mirror: lane i <-> 7-i inside each 8-lane segment
    e = min_f64(e, dpp_f64<0xB1>(e));      // quad_perm xor 1
    e = min_f64(e, dpp_f64<0x4E>(e));      // quad_perm xor 2
    return e;
}

// Per-bit metrics of one received value (convcode.py:575-587): m0 = cost of code bit 0, m1 = of bit 1.
__device__ __forceinline__ void bit_metrics(int type, double r, double &m0, double &m1) {
    if (type == CPX_VIT_HARD) {
        long long ri = (long long)r;            // r_codeword.astype(int) (:580)
        m0 = (double)(ri ^ 0ll);                // hamming_dist = sum of xor (utilities.py:130)
        m1 = (double)(ri ^ 1ll);
    } else if (type == CPX_VIT_SOFT) {
        double nll0 = fast_log<false>(exp(r) + 1.0);   // :582 (r is clipped to +-500: the argument is finite and >= 1)
        m0 = nll0;
        m1 = nll0 - r;                          // :583
    } else {
        double d0 = r - (-1.0), d1 = r - 1.0;   // i_codeword_array = 2*c - 1 (:586), euclid_dist utilities.py:152
        m0 = d0 * d0;
        m1 = d1 * d1;
    }
}

// LGS: log2(states) (S <= 64 lanes per codeword slot);  I_T: branches per state (2 or 4);
// SR: shift-register trellis (feed-forward, k = 1): predecessor j of state s is ((s << 1) & (S-1)) | j
//     and the input on every branch into s is s >> (LGS-1), so the traceback needs no table lookups;
// N_T: outputs per trellis step when known at compile time (2, 3), 0 = run-time n <= CPX_MAX_N.
// DM: the input is symbols to be hard-demodulated in the kernel (run-time n only, 'hard' metrics).
// NANX: the NaN-exact instantiation (redo launch only); the fast one returns whether a NaN was received ('soft').
// blk: the work item -- codewords blk * G .. blk * G + G - 1.
template <int LGS, int I_T, bool SR, int N_T, bool DM, bool NANX>
__device__ __forceinline__ bool wave_body(const VitParams &p, const int64_t blk, unsigned char *smem) {
    constexpr int PL = (I_T == 2) ? 1 : 2;          // decision bit planes
    constexpr int S = 1 << LGS, G = 64 >> LGS, CH = S;
    constexpr int NMAX = N_T ? N_T : CPX_MAX_N;
    const int lane = threadIdx.x;
    const int n = N_T ? N_T : p.n, NC = N_T ? (1 << N_T) : p.NC, k = p.k, RM = p.RS - 1;
    const int g = lane >> LGS, s = lane & (S - 1);

    double *bm = reinterpret_cast<double *>(smem);                              // [64][NC]
    double *pmbuf = bm + 64 * NC;                                               // [64] path metrics of the previous step
    double *mnbuf = pmbuf + 64;                                                 // [8] minima of the eight steps of a block (S = 64)
    unsigned long long *dring = reinterpret_cast<unsigned long long *>(mnbuf + 8);     // [RS][PL]
    unsigned short *ptab = reinterpret_cast<unsigned short *>(dring + (size_t)p.RS * PL);  // [S*I]
    unsigned char *bring = reinterpret_cast<unsigned char *>(ptab + S * I_T);   // [RS][G]

    const int64_t cw = blk * G + g;
    const bool valid_cw = cw < p.B;
    const double *x = p.coded + (valid_cw ? cw : 0) * p.len;

    // Predecessor metrics are exchanged through a 512-byte LDS buffer (one ds_write_b64 + ONE ds_read_b128
    // per step for shift-register trellises, whose two predecessors 2*(s mod S/2), +1 are adjacent):
    // 10 LDS cycles per step against 24 for the four ds_bpermute_b32 of a float64 shuffle pair
    // (measured 6.1 cycles per ds_bpermute_b32 per CU on MI355X, scripts/micro/bperm_bench.hip).
    int pidx[I_T], pcode[I_T];                               // predecessor slot in pmbuf / branch codeword
#pragma unroll
    for (int j = 0; j < I_T; j++) {
        pidx[j] = (g << LGS) + p.pred_state[s * I_T + j];
        pcode[j] = p.pred_code[s * I_T + j];
    }
    const double2 *ppair = reinterpret_cast<const double2 *>(pmbuf + (g << LGS) + 2 * (s & (S / 2 - 1)));
    if (!SR)
        for (int idx = lane; idx < S * I_T; idx += 64)
            ptab[idx] = (unsigned short)(p.pred_state[idx] | (p.pred_input[idx] << 8));

    double pm = (s == 0) ? 0.0 : __builtin_huge_val();      // path_metrics[:,0] = inf, [0][0] = 0 (:705-706)
    const int gshift = g * S;
    constexpr unsigned long long gmask = (S == 64) ? ~0ull : ((1ull << (S & 63)) - 1ull);
    int64_t next_out = 1;                                    // first output step not yet finalised

    // received values of the chunk being prepared (software prefetch one chunk ahead)
    double rcur[NMAX];
    const double2 *ys = DM ? p.ysym + (valid_cw ? cw : 0) * p.nsym : nullptr;
    auto load_chunk = [&](int64_t t_base, double *r) {
        const int64_t t = t_base + s;                        // lane (g, s) prepares step t of codeword g
        const bool have = valid_cw && (t <= p.Lk) && (t <= p.T);   // t > L//k -> padding (:722-734)
        if constexpr (DM) {
            // coded bit q of the codeword is bit q % nb (MSB first) of the label of symbol q / nb (modulation.py:121-123)
            unsigned last = ~0u;
            int label = 0;
#pragma unroll
            for (int j = 0; j < NMAX; j++) {
                double v = 0.0;
                if (have && j < n) {
                    const unsigned q = (unsigned)((t - 1) * n + j), sy = q / (unsigned)p.nb;
                    if (sy != last) { label = hard_label(p.cst, p.axes, p.M, p.nh, ys[sy]); last = sy; }
                    v = (double)((label >> (p.nb - 1 - (int)(q - sy * (unsigned)p.nb))) & 1);
                }
                r[j] = v;
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < NMAX; j++) {
            double v = (p.type == CPX_VIT_UNQUANTIZED) ? -1.0 : 0.0;
            if (have && j < n) v = x[(t - 1) * n + j];
            r[j] = v;
        }
    };
    load_chunk(1, rcur);
    unsigned long long nan_any = 0;                          // fast instantiation: lanes that received a NaN ('soft')
    bool poisoned_before = false, poisoned_now = false;      // NANX: see "NaN among 'soft' inputs"

    for (int64_t t_base = 1; t_base <= p.T; t_base += CH) {
        // ---------------- branch-metric table of this chunk ----------------
        {
            double m0[NMAX], m1[NMAX];
            bool nan_here = false;
#pragma unroll
            for (int j = 0; j < NMAX; j++) {
                double r = rcur[j];
                if (p.type == CPX_VIT_SOFT) {
                    nan_here |= (j < n) && (r != r);
                    r = fmin(fmax(r, -500.0), 500.0);                // coded_bits.clip(-500, 500) (:719); a NaN becomes -500
                }
                m0[j] = 0.0; m1[j] = 0.0;
                if (j < n) bit_metrics(p.type, r, m0[j], m1[j]);
            }
            if (!DM && p.type == CPX_VIT_SOFT) {                     // (a chunk is S steps: off the 'hard' / 'unquantized' paths)
                const unsigned long long nb = __ballot(nan_here);
                if constexpr (NANX) {
                    // lane (g, s) prepared step s of codeword g: the step is poisoned if the codeword was, or if a lane of its
                    // group at or below s saw a NaN
                    const unsigned long long nan_grp = (nb >> gshift) & gmask;
                    poisoned_now = poisoned_before || (nan_grp & ((2ull << s) - 1ull)) != 0;
                    poisoned_before = poisoned_before || nan_grp != 0;
                } else {
                    nan_any |= nb;
                }
            }
            double *row = bm + lane * NC;
            for (int c = 0; c < NC; c++) {
                double acc = 0.0;                            // NumPy add.reduce, n < 8: sequential from 0
#pragma unroll
                for (int j = 0; j < NMAX; j++)
                    if (j < n) acc += ((c >> (n - 1 - j)) & 1) ? m1[j] : m0[j];   // MSB-first bits (:622)
                row[c] = acc;
            }
        }
        if (t_base + CH <= p.T) load_chunk(t_base + CH, rcur);   // prefetch: lands during the forward loop
        __syncthreads();

        // ---------------- forward add-compare-select ----------------
        const int nsteps = (int)((p.T - t_base + 1 < CH) ? (p.T - t_base + 1) : CH);
        unsigned long long mydec0 = 0, mydec1 = 0;           // decision word(s) of step t_base + s, kept by lane (g, s)
        int mybest = 0;
        const double *rowbase = bm + (g << LGS) * NC;
        // one ACS step: updates pm, returns the decision ballot(s)
        auto acs = [&](int i, unsigned long long &w0, unsigned long long &w1) {
            const double *row = rowbase + i * NC;
            pmbuf[lane] = pm;                                              // LDS ops of one wave execute in order:
            asm volatile("" ::: "memory");                                 // the reads below see this step's metrics
            double best;
            w1 = 0;
            if (I_T == 2) {
                double a0, a1;
                if (SR) {
                    const double2 pp = *ppair;                             // both predecessors in one ds_read_b128
                    a0 = pp.x; a1 = pp.y;
                } else {
                    a0 = pmbuf[pidx[0]]; a1 = pmbuf[pidx[1]];
                }
                best = a0 + row[pcode[0]];                                 // pmetrics[0] (:629)
                const double c1 = a1 + row[pcode[1]];
                const bool d = c1 < best;                                  // first minimum wins (:633-642)
                best = d ? c1 : best;
                w0 = __ballot(d);
            } else {
                best = pmbuf[pidx[0]] + row[pcode[0]];
                int jb = 0;
#pragma unroll
                for (int j = 1; j < I_T; j++) {
                    const double c = pmbuf[pidx[j]] + row[pcode[j]];
                    if (c < best) { best = c; jb = j; }
                }
                w0 = __ballot(jb & 1);
                w1 = __ballot(jb & 2);
            }
            asm volatile("" ::: "memory");
            pm = best;
        };
        if constexpr (LGS == 6) {
            // S = 64: eight steps per block; their eight first-argmin reductions (:645) share one transposed tree.
            // A partial last block simply runs on (rows of steps > T hold padding metrics; their ring slots are never read).
            for (int i0 = 0; i0 < nsteps; i0 += 8) {
                double hist[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    unsigned long long w0, w1;
                    acs(i0 + u, w0, w1);
                    const bool mine = (s == i0 + u);         // (a v_writelane staging was measured: same speed)
                    mydec0 = mine ? w0 : mydec0;
                    if (PL == 2) mydec1 = mine ? w1 : mydec1;
                    hist[u] = pm;
                }
                const double e = min8_transposed(hist);
                // The minimum of step u sits in lanes 8*bitrev3(u)..+7.  All eight are handed to every lane through 64 bytes
                // of LDS (one masked store, four broadcast reads) instead of sixteen v_readlane; the eight first-argmin
                // states are found with scalar ballots / s_ff1, packed one byte each into an SGPR pair and dropped into
                // the lanes that own the steps with a single variable shift (a section profile showed this extraction,
                // done step by step, costing as much as the add-compare-select itself).
                if ((lane & 7) == 0) mnbuf[lane >> 3] = e;
                asm volatile("" ::: "memory");                 // in-order LDS: the reads below see the eight stores
                const double2 m01 = *reinterpret_cast<const double2 *>(mnbuf + 0), m23 = *reinterpret_cast<const double2 *>(mnbuf + 2);
                const double2 m45 = *reinterpret_cast<const double2 *>(mnbuf + 4), m67 = *reinterpret_cast<const double2 *>(mnbuf + 6);
                asm volatile("" ::: "memory");
                const double seg[8] = {m01.x, m01.y, m23.x, m23.y, m45.x, m45.y, m67.x, m67.y};
                unsigned long long pack = 0;                   // wave-uniform: first-argmin state of step i0+u in byte u
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    constexpr int SEG[8] = {0, 4, 2, 6, 1, 5, 3, 7};          // bitrev3(u): segment holding min(hist[u])
                    const unsigned long long eq = __ballot(hist[u] == seg[SEG[u]]);
                    const unsigned long long bst = eq ? (unsigned long long)(__ffsll((long long)eq) - 1) : 0ull;
                    pack |= bst << (8 * u);
                }
                const unsigned rel = (unsigned)(s - i0);       // this lane owns step i0 + rel when rel < 8
                mybest = (rel < 8u) ? (int)((pack >> ((rel & 7u) * 8u)) & 0xffull) : mybest;
            }
        } else {
            for (int i = 0; i < nsteps; i++) {
                unsigned long long w0, w1;
                acs(i, w0, w1);
                // first-argmin state of this step (:645)
                const double mn = group_min<LGS>(pm);
                const unsigned long long eq = __ballot(pm == mn);
                const unsigned long long grp = (eq >> gshift) & gmask;
                const int bst = grp ? (__ffsll((long long)grp) - 1) : 0;
                const bool mine = (s == i);
                mydec0 = mine ? w0 : mydec0;
                if (PL == 2) mydec1 = mine ? w1 : mydec1;
                mybest = mine ? bst : mybest;
            }
        }
        if constexpr (NANX) {
            // the decision word of a step is shared by the G codewords of the wavefront (bits [g S, g S + S) belong to codeword
            // g): clear the ranges of the codewords that are poisoned at this lane's step -- "first predecessor" everywhere
            const unsigned long long pb = __ballot(poisoned_now);
            if (pb) {
                // bit g S + s of pb: codeword g is poisoned at step s.  One bit per group, multiplied by the group mask, is S bits
                constexpr unsigned long long REP = (S == 64) ? 1ull : (~0ull / ((1ull << (S & 63)) - 1ull));   // 0x...0101 pattern
                const unsigned long long zero_bits = ((pb >> s) & REP) * gmask;
                mydec0 &= ~zero_bits;
                mydec1 &= ~zero_bits;
                if (poisoned_now) mybest = 0;                // argmin of all-NaN metrics (:645)
            }
        }
        {
            const int slot = (int)((t_base + s) & RM);       // slots of steps beyond T are never read
            dring[slot * PL] = mydec0;                        // identical words from the G codeword slots
            if (PL == 2) dring[slot * PL + 1] = mydec1;
            bring[slot * G + g] = (unsigned char)mybest;
        }
        __syncthreads();

        // ---------------- sliding traceback ----------------
        const int64_t t_done = t_base + nsteps - 1;
        const int64_t s_hi = (t_done >= p.T) ? p.T : (t_done - p.tb + 2);
        while (next_out <= s_hi) {
            const int64_t so = next_out + s;                 // output step owned by this lane
            if (so <= s_hi && valid_cw) {
                int64_t t0 = so + p.tb - 2;
                if (t0 > p.T) t0 = p.T;
                int st = bring[(int)(t0 & RM) * G + g];
                int sym;
                if (SR) {
                    for (int64_t tt = t0; tt > so; --tt) {
                        const int j = (int)((dring[(int)(tt & RM)] >> (gshift + st)) & 1ull);
                        st = ((st << 1) & (S - 1)) | j;
                    }
                    sym = st >> (LGS - 1);
                } else {
                    for (int64_t tt = t0; tt > so; --tt) {
                        const int slot = (int)(tt & RM);
                        int j = (int)((dring[slot * PL] >> (gshift + st)) & 1ull);
                        if (PL == 2) j |= (int)((dring[slot * PL + 1] >> (gshift + st)) & 1ull) << 1;
                        st = ptab[st * I_T + j] & 0xff;      // paths[current_state, j] (:651)
                    }
                    const int slot = (int)(so & RM);
                    int j = (int)((dring[slot * PL] >> (gshift + st)) & 1ull);
                    if (PL == 2) j |= (int)((dring[slot * PL + 1] >> (gshift + st)) & 1ull) << 1;
                    sym = ptab[st * I_T + j] >> 8;           // decoded_symbols[current_state, j] (:650)
                }
                for (int b = 0; b < k; b++) {
                    const int64_t pos = (so - 1) * k + b;
                    if (pos < p.L) p.bits[cw * p.L + pos] = (uint8_t)((sym >> (k - 1 - b)) & 1);   // dec2bitarray(sym, k) (:652)
                }
            }
            next_out = (next_out + CH <= s_hi + 1) ? next_out + CH : s_hi + 1;
        }
        __syncthreads();
    }
    return nan_any != 0;
}

template <int LGS, int I_T, bool SR, int N_T, bool DM = false>
__global__ __launch_bounds__(64) void viterbi_wave_kernel(VitParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const bool nan = wave_body<LGS, I_T, SR, N_T, DM, false>(p, (int64_t)blockIdx.x, smem);
    if (!DM && p.nanflags && threadIdx.x == 0) p.nanflags[blockIdx.x] = nan ? 1 : 0;
}

// Redo launch: wavefront w owns the work items 64 w .. 64 w + 63 of the launch it follows and re-decodes the flagged ones,
// one after the other, with the NaN-exact body.  N_T = 0 (run-time n): this kernel's speed does not matter.
template <int LGS, int I_T, bool SR>
__global__ __launch_bounds__(64) void viterbi_wave_redo_kernel(VitParams p, int64_t nitems) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int64_t item = (int64_t)blockIdx.x * 64 + threadIdx.x;
    unsigned long long todo = __ballot(item < nitems && p.nanflags[item] != 0);
    while (todo) {
        const int b = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        wave_body<LGS, I_T, SR, 0, false, true>(p, (int64_t)blockIdx.x * 64 + b, smem);
    }
}

// ---- S > 64 (total_memory 7: 128 states; total_memory 8, k = 1: 256 states): SPL = S/64 states per lane, one codeword per wavefront ----
// State q*64 + lane lives in register q of that lane.  Predecessor metrics go through a small LDS
// buffer (write all S metrics, read the I predecessors by address) instead of shuffles; everything
// else (branch-metric table, decision ring, sliding traceback, first-argmin rule) is as above.
// Table-driven traceback only.  The reference cannot build larger trellises (Trellis overflows int8
// for total_memory >= 8 on NumPy 2), so 128 states is the practical maximum.
template <int SPL, int I_T, bool NANX>
__device__ __forceinline__ bool wide_body(const VitParams &p, const int64_t cw, unsigned char *smem) {
    constexpr int PL = (I_T == 2) ? 1 : 2;
    constexpr int S = 64 * SPL, CH = 64;
    const int lane = threadIdx.x;
    const int n = p.n, NC = p.NC, k = p.k, RM = p.RS - 1;

    double *bm = reinterpret_cast<double *>(smem);                                   // [64][NC]
    double *pmbuf = bm + 64 * NC;                                                    // [S]
    unsigned long long *dring = reinterpret_cast<unsigned long long *>(pmbuf + S);   // [RS][PL][SPL]
    unsigned short *ptab = reinterpret_cast<unsigned short *>(dring + (size_t)p.RS * PL * SPL);   // [S*I]
    unsigned char *bring = reinterpret_cast<unsigned char *>(ptab + S * I_T);        // [RS]

    const double *x = p.coded + cw * p.len;
    int pst[SPL][I_T], pcode[SPL][I_T];
#pragma unroll
    for (int q = 0; q < SPL; q++)
#pragma unroll
        for (int j = 0; j < I_T; j++) {
            pst[q][j] = p.pred_state[(q * 64 + lane) * I_T + j];
            pcode[q][j] = p.pred_code[(q * 64 + lane) * I_T + j];
        }
    for (int idx = lane; idx < S * I_T; idx += 64)
        ptab[idx] = (unsigned short)(p.pred_state[idx] | (p.pred_input[idx] << 8));
    double pm[SPL];
#pragma unroll
    for (int q = 0; q < SPL; q++) pm[q] = (q == 0 && lane == 0) ? 0.0 : __builtin_huge_val();
    int64_t next_out = 1;
    unsigned long long nan_any = 0;                                // fast instantiation: a NaN was received ('soft')
    bool poisoned = false;                                         // NANX: a NaN was received in an earlier chunk (see wave_body)
    unsigned long long nan_steps = 0;

    for (int64_t t_base = 1; t_base <= p.T; t_base += CH) {
        {   // branch-metric table: lane i prepares step t_base + i
            const int64_t t = t_base + lane;
            const bool have = (t <= p.Lk) && (t <= p.T);
            double m0[CPX_MAX_N], m1[CPX_MAX_N];
            bool nan_here = false;
#pragma unroll
            for (int j = 0; j < CPX_MAX_N; j++) {
                double r = (p.type == CPX_VIT_UNQUANTIZED) ? -1.0 : 0.0;
                if (have && j < n) r = x[(t - 1) * n + j];
                if (p.type == CPX_VIT_SOFT) {
                    nan_here |= (j < n) && (r != r);
                    r = fmin(fmax(r, -500.0), 500.0);
                }
                m0[j] = 0.0; m1[j] = 0.0;
                if (j < n) bit_metrics(p.type, r, m0[j], m1[j]);
            }
            if (p.type == CPX_VIT_SOFT) {
                nan_steps = __ballot(nan_here);                   // bit i: step t_base + i received a NaN
                nan_any |= nan_steps;
            }
            for (int c = 0; c < NC; c++) {
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < CPX_MAX_N; j++)
                    if (j < n) acc += ((c >> (n - 1 - j)) & 1) ? m1[j] : m0[j];
                bm[lane * NC + c] = acc;
            }
        }
        __syncthreads();
        const int nsteps = (int)((p.T - t_base + 1 < CH) ? (p.T - t_base + 1) : CH);
        for (int i = 0; i < nsteps; i++) {
            const int64_t t = t_base + i;
            const double *row = bm + i * NC;
#pragma unroll
            for (int q = 0; q < SPL; q++) pmbuf[q * 64 + lane] = pm[q];
            __syncthreads();
            const bool pz = NANX && (poisoned || (nan_steps & ((2ull << i) - 1ull)) != 0);   // wave-uniform: step t is poisoned
            unsigned long long w[SPL][2];
#pragma unroll
            for (int q = 0; q < SPL; q++) {
                double best = pmbuf[pst[q][0]] + row[pcode[q][0]];
                int jb = 0;
#pragma unroll
                for (int j = 1; j < I_T; j++) {
                    const double c = pmbuf[pst[q][j]] + row[pcode[q][j]];
                    if (c < best) { best = c; jb = j; }                        // first minimum wins
                }
                pm[q] = best;
                w[q][0] = pz ? 0ull : __ballot(jb & 1);
                w[q][1] = (PL == 2 && !pz) ? __ballot(jb & 2) : 0ull;
            }
            double mn = pm[0];
#pragma unroll
            for (int q = 1; q < SPL; q++) mn = (pm[q] < mn) ? pm[q] : mn;
            mn = group_min<6>(mn);
            int bst = -1;
#pragma unroll
            for (int q = 0; q < SPL; q++) {
                const unsigned long long eq = __ballot(pm[q] == mn);
                if (bst < 0 && eq) bst = q * 64 + __ffsll((long long)eq) - 1;   // lowest state index among ties
            }
            const int slot = (int)(t & RM);
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < SPL; q++) {
                    dring[(slot * PL) * SPL + q] = w[q][0];
                    if (PL == 2) dring[(slot * PL + 1) * SPL + q] = w[q][1];
                }
                bring[slot] = (unsigned char)((bst < 0 || pz) ? 0 : bst);
            }
            __syncthreads();
        }
        poisoned = poisoned || nan_steps != 0;
        const int64_t t_done = t_base + nsteps - 1;
        const int64_t s_hi = (t_done >= p.T) ? p.T : (t_done - p.tb + 2);
        auto decision = [&](int64_t tt, int st) {
            const int slot = (int)(tt & RM);
            int j = (int)((dring[(slot * PL) * SPL + (st >> 6)] >> (st & 63)) & 1ull);
            if (PL == 2) j |= (int)((dring[(slot * PL + 1) * SPL + (st >> 6)] >> (st & 63)) & 1ull) << 1;
            return j;
        };
        while (next_out <= s_hi) {
            const int64_t so = next_out + lane;
            if (so <= s_hi) {
                int64_t t0 = so + p.tb - 2;
                if (t0 > p.T) t0 = p.T;
                int st = bring[(int)(t0 & RM)];
                for (int64_t tt = t0; tt > so; --tt) st = ptab[st * I_T + decision(tt, st)] & 0xff;
                const int sym = ptab[st * I_T + decision(so, st)] >> 8;
                for (int b = 0; b < k; b++) {
                    const int64_t pos = (so - 1) * k + b;
                    if (pos < p.L) p.bits[cw * p.L + pos] = (uint8_t)((sym >> (k - 1 - b)) & 1);
                }
            }
            next_out = (next_out + CH <= s_hi + 1) ? next_out + CH : s_hi + 1;
        }
        __syncthreads();
    }
    return nan_any != 0;
}

template <int SPL, int I_T>
__global__ __launch_bounds__(64) void viterbi_wide_kernel(VitParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const bool nan = wide_body<SPL, I_T, false>(p, (int64_t)blockIdx.x, smem);
    if (p.nanflags && threadIdx.x == 0) p.nanflags[blockIdx.x] = nan ? 1 : 0;
}

template <int SPL, int I_T>
__global__ __launch_bounds__(64) void viterbi_wide_redo_kernel(VitParams p, int64_t nitems) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int64_t item = (int64_t)blockIdx.x * 64 + threadIdx.x;
    unsigned long long todo = __ballot(item < nitems && p.nanflags[item] != 0);
    while (todo) {
        const int b = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        wide_body<SPL, I_T, true>(p, (int64_t)blockIdx.x * 64 + b, smem);
    }
}

int next_pow2(int v) {
    int r = 1;
    while (r < v) r <<= 1;
    return r;
}

}  // namespace

extern "C" {

int cpx_trellis_create(int k, int n, int n_states, int n_inputs, const int32_t *next_state_table,
                       const int32_t *output_table, cpx_trellis **out) {
    CPX_REQUIRE(out && next_state_table && output_table, CPX_EINVAL, "cpx_trellis_create: null pointer");
    CPX_REQUIRE(k >= 1 && n >= 1 && n_inputs == (1 << k), CPX_EINVAL, "cpx_trellis_create: n_inputs must be 2^k");
    CPX_REQUIRE(n_states >= 1 && (n_states & (n_states - 1)) == 0, CPX_EINVAL,
                "cpx_trellis_create: n_states must be a power of two");
    CPX_REQUIRE(n_states <= CPX_MAX_STATES, CPX_ELIMIT, "cpx_trellis_create: at most %d states supported", CPX_MAX_STATES);
    CPX_REQUIRE(n <= 16, CPX_ELIMIT, "cpx_trellis_create: n <= 16");
    int rc = ensure_device();
    if (rc) return rc;
    const int S = n_states, I = n_inputs;
    for (int i = 0; i < S * I; i++) {
        CPX_REQUIRE(next_state_table[i] >= 0 && next_state_table[i] < S, CPX_EINVAL, "next_state_table entry out of range");
        CPX_REQUIRE(output_table[i] >= 0 && output_table[i] < (1 << n), CPX_EINVAL, "output_table entry out of range");
    }
    cpx_trellis *t = new cpx_trellis;
    t->k = k; t->n = n; t->S = S; t->I = I;
    t->next_state.assign(next_state_table, next_state_table + S * I);
    t->output.assign(output_table, output_table + S * I);
    // predecessor lists in np.where (row-major) order: convcode.py:561-572
    t->pred_state.assign(S * I, -1); t->pred_input.assign(S * I, -1); t->pred_code.assign(S * I, 0);
    bool regular = true;
    std::vector<int> cnt(S, 0);
    for (int ps = 0; ps < S && regular; ps++)
        for (int i = 0; i < I; i++) {
            int ns = next_state_table[ps * I + i];
            if (cnt[ns] >= I) { regular = false; break; }
            t->pred_state[ns * I + cnt[ns]] = ps;
            t->pred_input[ns * I + cnt[ns]] = i;
            t->pred_code[ns * I + cnt[ns]] = output_table[ps * I + i];
            cnt[ns]++;
        }
    for (int s2 = 0; s2 < S; s2++) if (cnt[s2] != I) regular = false;
    if (!regular) {
        delete t;
        set_error("cpx_trellis_create: every state needs exactly %d incoming branches (the reference indexes "
                  "pmetrics[number_inputs], convcode.py:604-629)", I);
        return CPX_EINVAL;
    }
    (void)hipGetDevice(&t->device);
    size_t bytes = sizeof(int32_t) * S * I;
    int32_t **dst[5] = {&t->d_next, &t->d_out, &t->d_pred_state, &t->d_pred_input, &t->d_pred_code};
    const int32_t *src[5] = {t->next_state.data(), t->output.data(), t->pred_state.data(), t->pred_input.data(),
                             t->pred_code.data()};
    for (int i = 0; i < 5; i++) {
        CPX_HIP(hipMalloc((void **)dst[i], bytes));
        CPX_HIP(hipMemcpy(*dst[i], src[i], bytes, hipMemcpyHostToDevice));
    }
    *out = t;
    return CPX_OK;
}

int cpx_trellis_destroy(cpx_trellis *t) {
    if (!t) return CPX_OK;
    if (t->spec_mod) (void)hipModuleUnload(t->spec_mod);
    (void)hipFree(t->d_next); (void)hipFree(t->d_out);
    (void)hipFree(t->d_pred_state); (void)hipFree(t->d_pred_input); (void)hipFree(t->d_pred_code);
    delete t;
    return CPX_OK;
}

// shift-register structure (feed-forward, k = 1): predecessor j of state s is ((s << 1) & (S - 1)) | j, input s >> (lgS - 1)
static bool shift_register(const cpx_trellis *t) {
    int lgS = 0;
    while ((1 << lgS) < t->S) lgS++;
    if (!(t->I == 2 && t->k == 1 && lgS >= 1)) return false;
    for (int s2 = 0; s2 < t->S; s2++)
        for (int j = 0; j < 2; j++)
            if (t->pred_state[s2 * 2 + j] != (((s2 << 1) & (t->S - 1)) | j) || t->pred_input[s2 * 2 + j] != (s2 >> (lgS - 1)))
                return false;
    return true;
}

// ring size and dynamic LDS of the state-per-lane kernels for this trellis and p.tb (sets p.RS)
static size_t wave_lds_bytes(const cpx_trellis *t, VitParams &p, size_t *lds) {
    const int PL = (t->I == 2) ? 1 : 2;
    if (t->S > 64) {
        const int SPL = t->S / 64;
        p.RS = next_pow2(64 + p.tb);
        *lds = sizeof(double) * 64 * p.NC + sizeof(double) * t->S + sizeof(unsigned long long) * p.RS * PL * SPL +
               sizeof(unsigned short) * t->S * t->I + (size_t)p.RS;
    } else {
        const int S = t->S, G = 64 / S, CH = S;
        p.RS = next_pow2(CH + p.tb);
        *lds = sizeof(double) * 64 * p.NC + sizeof(double) * (64 + 8) + sizeof(unsigned long long) * p.RS * PL +
               sizeof(unsigned short) * S * t->I + (size_t)p.RS * G;
    }
    return *lds;
}
static int wave_lds(const cpx_trellis *t, VitParams &p, size_t *lds) {
    CPX_REQUIRE(wave_lds_bytes(t, p, lds) <= 64 * 1024, CPX_ELIMIT, "viterbi: tb_depth %d needs %zu B of LDS (> 64 KiB)", p.tb, *lds);
    return CPX_OK;
}

// The redo launch behind a 'soft' decode (see "NaN among 'soft' inputs"): `nitems` flag bytes at p.nanflags, one per work
// item of the state-per-lane body for this trellis (64 / S codewords; one codeword for 64 and 128 states).
static int launch_redo(const cpx_trellis *t, VitParams p, int64_t nitems, hipStream_t st) {
    size_t lds = 0;
    if (int rcl = wave_lds(t, p, &lds)) return rcl;
    const int64_t nw = (nitems + 63) / 64;
    CPX_REQUIRE(nw < (1ll << 31), CPX_ELIMIT, "viterbi: batch too large");
    const dim3 grid((unsigned)nw), block(64);
    if (t->S > 64) {
        if (t->S == 256) hipLaunchKernelGGL((viterbi_wide_redo_kernel<4, 2>), grid, block, lds, st, p, nitems);
        else if (t->I == 2) hipLaunchKernelGGL((viterbi_wide_redo_kernel<2, 2>), grid, block, lds, st, p, nitems);
        else hipLaunchKernelGGL((viterbi_wide_redo_kernel<2, 4>), grid, block, lds, st, p, nitems);
    } else {
        const bool sr = shift_register(t);
#define REDO_CASE(LG)                                                                                              \
    case LG:                                                                                                       \
        if (t->I == 4) hipLaunchKernelGGL((viterbi_wave_redo_kernel<LG, 4, false>), grid, block, lds, st, p, nitems);   \
        else if (sr) hipLaunchKernelGGL((viterbi_wave_redo_kernel<LG, 2, true>), grid, block, lds, st, p, nitems);      \
        else hipLaunchKernelGGL((viterbi_wave_redo_kernel<LG, 2, false>), grid, block, lds, st, p, nitems);             \
        break;
        switch (p.lgS) {
            REDO_CASE(1) REDO_CASE(2) REDO_CASE(3) REDO_CASE(4) REDO_CASE(5) REDO_CASE(6)
            default: set_error("viterbi: unsupported number of states %d", t->S); return CPX_ELIMIT;
        }
#undef REDO_CASE
    }
    CPX_HIP(hipGetLastError());
    return CPX_OK;
}

struct DemodSrc {                  // fused hard demodulation: symbols instead of decoder input
    const cpx_modem *m;
    const double *d_y;             // [B][nsym][2]
    int64_t nsym;
};

// ---- side stream of the remainder (viterbi_dispatch, round 6) -------------------------------------------------------------------
static bool overlap_enabled() {
    static const bool on = [] { const char *e = getenv("CPX_VITERBI_OVERLAP"); return !(e && e[0] == '0'); }();
    return on;
}

// One lowest-priority stream per device, shared by the host threads (its queue orders their remainders); the fork / join events are
// per THREAD and device: a shared pair would let thread B's record slip between thread A's record and A's hipStreamWaitEvent, and A's
// remainder would then wait for B's stream instead of its own inputs.  (An event may be re-recorded while an earlier wait on it is
// pending: the wait took the record that was current when it was issued.  A thread's pair lives as long as the process.)
static int side_stream(hipStream_t *st, hipEvent_t *fork, hipEvent_t *join) {
    static std::mutex mu;
    static hipStream_t s_side[64] = {};
    constexpr int PAIRS = 4;                                      // consecutive calls of a thread take different pairs (belt and braces:
    static thread_local hipEvent_t tl_ev[64][2 * PAIRS] = {};     //  a re-record never touches an event whose wait was issued one call ago)
    static thread_local unsigned tl_next[64] = {};
    int dev = 0;
    CPX_HIP(hipGetDevice(&dev));
    CPX_REQUIRE(dev >= 0 && dev < 64, CPX_ELIMIT, "viterbi: device index %d", dev);
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!s_side[dev]) {
            int least = 0, greatest = 0;
            CPX_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
            CPX_HIP(hipStreamCreateWithPriority(&s_side[dev], hipStreamNonBlocking, least));
        }
    }
    const unsigned k = 2 * (tl_next[dev]++ % PAIRS);
    if (!tl_ev[dev][k]) {
        CPX_HIP(hipEventCreateWithFlags(&tl_ev[dev][k], hipEventDisableTiming));
        CPX_HIP(hipEventCreateWithFlags(&tl_ev[dev][k + 1], hipEventDisableTiming));
    }
    *st = s_side[dev]; *fork = tl_ev[dev][k]; *join = tl_ev[dev][k + 1];
    return CPX_OK;
}

static int viterbi_dispatch(const cpx_trellis *t, const double *d_coded, const DemodSrc *dm, int64_t B, int64_t len, int64_t L,
                            int64_t n_steps, int tb_depth, int decoding_type, uint8_t *d_bits, void *stream) {
    CPX_REQUIRE(t, CPX_EINVAL, "viterbi: null trellis");
    if (int rcd = check_handle_device(t->device, "viterbi")) return rcd;
    CPX_REQUIRE(decoding_type >= 0 && decoding_type <= 2, CPX_EINVAL,
                "The available decoding types are \"hard\", \"soft\" and \"unquantized");
    CPX_REQUIRE(B >= 0 && len >= 0 && L >= 0, CPX_EINVAL, "viterbi: negative size");
    CPX_REQUIRE(tb_depth >= 2, CPX_EINVAL, "viterbi: tb_depth must be >= 2");
    CPX_REQUIRE((L / t->k) * (int64_t)t->n <= len, CPX_EINVAL, "viterbi: L inconsistent with len");
    // what the specialised kernels below are instantiated for; every other trellis / window takes viterbi_generic.hip
    // (256 states with k = 1 -- K = 9, round 4 -- run four states per lane on the wide kernel: its tables hold 8-bit state numbers)
    bool general = !(t->I == 2 || t->I == 4) || t->n > CPX_MAX_N || t->S < 2 || (t->S > 128 && !(t->S == 256 && t->I == 2));
    if (!general) {
        VitParams q;
        q.tb = tb_depth; q.NC = 1 << t->n;
        size_t need = 0;
        general = wave_lds_bytes(t, q, &need) > 64 * 1024;       // the traceback ring of the state-per-lane kernels lives in LDS
    }
    if (dm) {
        CPX_REQUIRE(!general, CPX_ELIMIT, "demod_hard_viterbi: this trellis / traceback depth takes the two-call path (demodulate, then viterbi_decode)");
    }
    if (B == 0 || L == 0) return CPX_OK;
    hipStream_t st = pick_stream(stream);
    if (n_steps <= 0 || n_steps * t->k < L) CPX_HIP(hipMemsetAsync(d_bits, 0, (size_t)(B * L), st));
    if (n_steps <= 0) return CPX_OK;
    note_kernel("");
    // (a forced 'general' path does not apply to the fused hard-demodulation entry point: it hands symbols in `dm`, d_coded is null)
    if (general || (!dm && (viterbi_path_flags() & 16)))
        return viterbi_generic(t, d_coded, B, len, L, n_steps, tb_depth, decoding_type, d_bits, st);
    // 'soft': one flag byte per work item of the launches below (see "NaN among 'soft' inputs"); scratch-arena slot 3
    uint8_t *nanflags = nullptr;
    if (decoding_type == CPX_VIT_SOFT && !dm) {
        void *w = nullptr;
        if (int rcw = workspace(st, 3, (size_t)B + 64, &w)) return rcw;
        nanflags = static_cast<uint8_t *>(w);
    }
    // parameters of the state-per-lane kernels for `nb` codewords starting at (`coded`, `bits`); `flags`: their flag bytes
    auto wave_params = [&](VitParams &p, const double *coded, uint8_t *bits, int64_t nb, uint8_t *flags) {
        p.coded = coded; p.bits = bits;
        p.pred_state = t->d_pred_state; p.pred_input = t->d_pred_input; p.pred_code = t->d_pred_code;
        p.B = nb; p.len = len; p.L = L; p.T = n_steps; p.Lk = L / t->k;
        p.k = t->k; p.n = t->n; p.I = t->I; p.NC = 1 << t->n; p.type = decoding_type; p.tb = tb_depth;
        p.ysym = nullptr; p.cst = nullptr; p.axes = nullptr; p.nsym = 0; p.M = 0; p.nb = 1; p.nh = 0;
        p.nanflags = flags;
        int lg = 0;
        while ((1 << lg) < t->S) lg++;
        p.lgS = lg;
    };
    hipStream_t st_rem = st;                                     // the stream of the state-per-lane launches below
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool overlapped = false;
    auto join = [&](int rc) {                                    // every exit below the fork: `st` continues behind the side stream
        if (overlapped && (hipEventRecord(ev_join, st_rem) != hipSuccess || hipStreamWaitEvent(st, ev_join, 0) != hipSuccess)) {
            (void)hipStreamSynchronize(st_rem);
            set_error("viterbi: joining the side stream failed");
            return rc ? rc : CPX_EHIP;
        }
        return rc;
    };
    if (!dm) {   // large batches of the standard rate-1/2 codes: one codeword per lane (viterbi_cw.hip).  That path runs in rounds
        // of one wavefront of 64 codewords per SIMD, each as long as a full one; a last round that would fill less than 45 % of
        // the chip is cheaper on the wave kernels below, whose time is proportional to the batch (config 2: 54 us per 1000
        // codewords against 1.55 ms per round, break-even at 0.44 of a round -- scripts/micro/split_probe.py: 30 000
        // codewords 1.63 ms on the wave kernels, 1.53 ms as a round; 99 536: 3.59 ms as round + wave kernels, 3.01 as two rounds)
        const int64_t round = (int64_t)device_cus() * 4 * 64;
        int64_t Bcw = B;
        if (!(viterbi_path_flags() & 2) && B > round && 20 * (B % round) < 9 * round) Bcw = B / round * round;
        // Round 6: the remainder runs BESIDE the rounds.  The rounds take the 32-slot ring stored once (launch_fused_lean: 93 instead of
        // 157 KB of LDS per workgroup, priority 3), which leaves the remainder's state-per-lane waves room on every CU; they are issued on a
        // lowest-priority side stream, forked from `st` before the rounds are launched (same inputs) and joined behind them, so that the
        // dispatcher places the round's 256 workgroups first -- one per CU -- and the remainder around them.  'soft', default depth, float64,
        // built-in 64-state pairs only (where the lean flavour exists); CPX_VITERBI_OVERLAP=0 keeps one stream and the mirrored ring.
        hipStream_t st_side = nullptr;
        bool armed = false;
        if (Bcw < B && overlap_enabled() && t->S == 64 && decoding_type == CPX_VIT_SOFT && tb_depth == 30 && !precision_fast()) {
            armed = side_stream(&st_side, &ev_fork, &ev_join) == CPX_OK && hipEventRecord(ev_fork, st) == hipSuccess;
            if (!armed) (void)hipGetLastError();
        }
        int rc_cw = CPX_OK;
        viterbi_lean_ring(armed);
        const bool took_cw = viterbi_codeword_path(t, d_coded, Bcw, len, L, n_steps, tb_depth, decoding_type, d_bits, nanflags, st, &rc_cw);
        viterbi_lean_ring(false);
        if (took_cw) {
            if (rc_cw != CPX_OK) return rc_cw;
            if (nanflags) {
                // codeword path: one flag per item of the redo kernel -- 64 / S consecutive codewords, the codewords of one
                // wavefront of the state-per-lane kernel (one codeword for 64 states)
                const int64_t items = (Bcw + (64 / t->S) - 1) / (64 / t->S);
                VitParams q;
                wave_params(q, d_coded, d_bits, Bcw, nanflags);
                if (int rcr = launch_redo(t, q, items, st)) return rcr;
                nanflags += items;
            }
            if (Bcw == B) return CPX_OK;
            if (armed && strstr(last_kernel_name(), "ring stored once")) {   // (the rounds are in `st`'s queue already: dispatched first)
                if (hipStreamWaitEvent(st_side, ev_fork, 0) == hipSuccess) { st_rem = st_side; overlapped = true; }
                else (void)hipGetLastError();
            }
            d_coded += Bcw * len;
            d_bits += Bcw * L;
            B -= Bcw;
        }
    }

    VitParams p;
    wave_params(p, d_coded, d_bits, B, nanflags);
    if (dm) {
        p.ysym = reinterpret_cast<const double2 *>(dm->d_y);
        p.cst = reinterpret_cast<const double2 *>(dm->m->d_const);
        p.axes = dm->m->separable ? dm->m->d_axes : nullptr;
        p.nsym = dm->nsym; p.M = dm->m->M; p.nb = dm->m->nbits; p.nh = dm->m->separable ? dm->m->nbits / 2 : 0;
        CPX_REQUIRE(t->S <= 64, CPX_ELIMIT, "demod_hard_viterbi: trellises above 64 states take the two-call path");
    }
    const int lgS = p.lgS;
    size_t lds = 0;
    if (int rcl = wave_lds(t, p, &lds)) return join(rcl);
    if (t->S > 64) {                                             // 128 states: two states per lane
        if (B >= (1ll << 31)) { set_error("viterbi: batch too large"); return join(CPX_ELIMIT); }
        if (t->S == 256) hipLaunchKernelGGL((viterbi_wide_kernel<4, 2>), dim3((unsigned)B), dim3(64), lds, st_rem, p);
        else if (t->I == 2) hipLaunchKernelGGL((viterbi_wide_kernel<2, 2>), dim3((unsigned)B), dim3(64), lds, st_rem, p);
        else hipLaunchKernelGGL((viterbi_wide_kernel<2, 4>), dim3((unsigned)B), dim3(64), lds, st_rem, p);
        CPX_HIP(hipGetLastError());
        if (p.nanflags) if (int rcr = launch_redo(t, p, B, st_rem)) return join(rcr);
        note_kernel("viterbi_wide_kernel<%d,%d>", t->S / 64, t->I);
        return join(CPX_OK);
    }
    const int S = t->S, G = 64 / S;
    const int64_t nblocks = (B + G - 1) / G;
    if (nblocks >= (1ll << 31)) { set_error("viterbi: batch too large"); return join(CPX_ELIMIT); }
    dim3 grid((unsigned)nblocks), block(64);
    const bool sr = shift_register(t);                       // => arithmetic traceback (no predecessor table lookups)
#define VIT_LAUNCH(LG, IT, SRV)                                                                             \
    do {                                                                                                    \
        if (dm) hipLaunchKernelGGL((viterbi_wave_kernel<LG, IT, SRV, 0, true>), grid, block, lds, st_rem, p);    \
        else if (t->n == 2) hipLaunchKernelGGL((viterbi_wave_kernel<LG, IT, SRV, 2>), grid, block, lds, st_rem, p);    \
        else if (t->n == 3) hipLaunchKernelGGL((viterbi_wave_kernel<LG, IT, SRV, 3>), grid, block, lds, st_rem, p); \
        else hipLaunchKernelGGL((viterbi_wave_kernel<LG, IT, SRV, 0>), grid, block, lds, st_rem, p);             \
    } while (0)
#define VIT_CASE(LG)                                       \
    case LG:                                               \
        if (t->I == 4) VIT_LAUNCH(LG, 4, false);           \
        else if (sr) VIT_LAUNCH(LG, 2, true);              \
        else VIT_LAUNCH(LG, 2, false);                     \
        break;
    switch (lgS) {
        VIT_CASE(1) VIT_CASE(2) VIT_CASE(3) VIT_CASE(4) VIT_CASE(5) VIT_CASE(6)
        default: set_error("viterbi: unsupported number of states %d", S); return join(CPX_ELIMIT);
    }
#undef VIT_CASE
#undef VIT_LAUNCH
    CPX_HIP(hipGetLastError());
    if (p.nanflags) if (int rcr = launch_redo(t, p, nblocks, st_rem)) return join(rcr);
    {
        char first[160];                                         // a leading round on the codeword path, if any
        snprintf(first, sizeof(first), "%s", last_kernel_name());
        note_kernel("%s%sviterbi_wave_kernel<%d,%d,%s,%d%s>%s", first, first[0] ? " + " : "", lgS, t->I,
                    (t->I == 2 && sr) ? "true" : "false", (!dm && (t->n == 2 || t->n == 3)) ? t->n : 0, dm ? ",demod" : "",
                    overlapped ? " (beside the round, side stream)" : "");
    }
    return join(CPX_OK);
}

int cpx_viterbi_decode_batch_dev(const cpx_trellis *t, const double *d_coded, int64_t B, int64_t len, int64_t L,
                                 int64_t n_steps, int tb_depth, int decoding_type, uint8_t *d_bits, void *stream) {
    CPX_TRACE("cpx_viterbi_decode_batch_dev");
    cpx::IssueGuard issue_guard;
    return viterbi_dispatch(t, d_coded, nullptr, B, len, L, n_steps, tb_depth, decoding_type, d_bits, stream);
}

int cpx_demod_hard_viterbi_batch_dev(const cpx_modem *m, const cpx_trellis *t, const double *d_y_re_im, int64_t B,
                                     int64_t nsym, int64_t L, int64_t n_steps, int tb_depth, uint8_t *d_bits, void *stream) {
    CPX_TRACE("cpx_demod_hard_viterbi_batch_dev");
    cpx::IssueGuard issue_guard;
    CPX_REQUIRE(m && t, CPX_EINVAL, "demod_hard_viterbi: null handle");
    if (int rcd = check_handle_device(m->device, "demod_hard_viterbi")) return rcd;
    CPX_REQUIRE(B >= 0 && nsym >= 0, CPX_EINVAL, "demod_hard_viterbi: negative size");
    CPX_REQUIRE(nsym * (int64_t)m->nbits < (1ll << 31), CPX_ELIMIT, "demod_hard_viterbi: codeword too long");
    DemodSrc dm{m, d_y_re_im, nsym};
    // len = the coded bits one codeword's symbols carry, exactly what demodulate(y, 'hard') would have returned
    return viterbi_dispatch(t, nullptr, &dm, B, nsym * m->nbits, L, n_steps, tb_depth, CPX_VIT_HARD, d_bits, stream);
}

int cpx_demod_hard_viterbi_batch(const cpx_modem *m, const cpx_trellis *t, const double *y_re_im, int64_t B, int64_t nsym,
                                 int64_t L, int64_t n_steps, int tb_depth, uint8_t *bits) {
    CPX_TRACE("cpx_demod_hard_viterbi_batch");
    CPX_REQUIRE(m && t && (y_re_im || B * nsym == 0) && (bits || B * L == 0), CPX_EINVAL, "demod_hard_viterbi: null pointer");
    int rc = ensure_device();
    if (rc) return rc;
    if (B == 0 || L == 0) return CPX_OK;
    DevBuf din, dout;
    if ((rc = din.alloc(sizeof(double) * 2 * (size_t)(B * nsym)))) return rc;
    if ((rc = dout.alloc((size_t)(B * L)))) return rc;
    hipStream_t st = lib_stream();
    CPX_HIP(hipMemcpyAsync(din.p, y_re_im, sizeof(double) * 2 * (size_t)(B * nsym), hipMemcpyHostToDevice, st));
    rc = cpx_demod_hard_viterbi_batch_dev(m, t, din.as<double>(), B, nsym, L, n_steps, tb_depth, dout.as<uint8_t>(), st);
    if (rc) return rc;
    CPX_HIP(hipMemcpyAsync(bits, dout.p, (size_t)(B * L), hipMemcpyDeviceToHost, st));
    CPX_HIP(hipStreamSynchronize(st));
    return CPX_OK;
}

int cpx_viterbi_decode_batch(const cpx_trellis *t, const double *coded, int64_t B, int64_t len, int64_t L,
                             int64_t n_steps, int tb_depth, int decoding_type, uint8_t *bits) {
    CPX_TRACE("cpx_viterbi_decode_batch");
    CPX_REQUIRE(t && (coded || B * len == 0) && (bits || B * L == 0), CPX_EINVAL, "viterbi: null pointer");
    int rc = ensure_device();
    if (rc) return rc;
    if (B == 0 || L == 0) return CPX_OK;
    DevBuf din, dout;
    if ((rc = din.alloc(sizeof(double) * (size_t)(B * len)))) return rc;
    if ((rc = dout.alloc((size_t)(B * L)))) return rc;
    hipStream_t st = lib_stream();
    CPX_HIP(hipMemcpyAsync(din.p, coded, sizeof(double) * (size_t)(B * len), hipMemcpyHostToDevice, st));
    rc = cpx_viterbi_decode_batch_dev(t, din.as<double>(), B, len, L, n_steps, tb_depth, decoding_type,
                                      dout.as<uint8_t>(), st);
    if (rc) return rc;
    CPX_HIP(hipMemcpyAsync(bits, dout.p, (size_t)(B * L), hipMemcpyDeviceToHost, st));
    CPX_HIP(hipStreamSynchronize(st));
    return CPX_OK;
}

int cpx_viterbi_decode_batch_i64(const cpx_trellis *t, const double *coded, int64_t B, int64_t len, int64_t L,
                                 int64_t n_steps, int tb_depth, int decoding_type, int64_t *bits64) {
    CPX_TRACE("cpx_viterbi_decode_batch_i64");
    CPX_REQUIRE(t && (coded || B * len == 0) && (bits64 || B * L == 0), CPX_EINVAL, "viterbi: null pointer");
    int rc = ensure_device();
    if (rc) return rc;
    if (B == 0 || L == 0) return CPX_OK;
    const size_t nout = (size_t)(B * L);
    // compact bits cross PCIe into a pinned staging block and are widened by host threads straight into the caller's
    // array (first touch of its pages included): 8x less PCIe traffic than widening on the device, and the page
    // faults of a fresh 8-byte-per-bit result array are spread over the cores
    static std::mutex mu;
    static uint8_t *stage = nullptr;
    static size_t stage_cap = 0;
    std::lock_guard<std::mutex> lk(mu);
    // the arena blocks below are touched from this function's own three streams: hold the device's issue lock for the whole
    // call, so that cpx_release_workspace (which takes every issue lock, then synchronises the library stream only) cannot
    // free them while copies and kernels of the pipeline are still in flight
    cpx::IssueGuard issue_guard;
    if (stage_cap < nout) {
        if (stage) (void)hipHostFree(stage);
        stage = nullptr; stage_cap = 0;
        CPX_HIP(hipHostMalloc((void **)&stage, nout, hipHostMallocDefault));
        stage_cap = nout;
    }
    // device staging from the scratch arena -- slots 6 / 7 of the library stream, used by nothing else (this function is
    // serialised by `mu`; the blocks are touched from the pipeline's own streams, so they must not be shared with entry points
    // that rely on the library stream's order): a
    // hipMalloc + hipFree of 1.08 GB per call is about a millisecond of a 23 ms call
    ArenaBuf din, dout;
    if ((rc = workspace(lib_stream(), 6, sizeof(double) * (size_t)(B * len), &din.p))) return rc;
    if ((rc = workspace(lib_stream(), 7, nout, &dout.p))) return rc;
    unsigned nt = std::thread::hardware_concurrency();
    if (nt > 16) nt = 16;
    if (nt < 1) nt = 1;
    auto widen = [&](size_t lo_all, size_t hi_all, unsigned threads) {     // bits64[lo_all, hi_all) = stage[...]
        auto work = [&](unsigned i) {
            const size_t n = hi_all - lo_all, lo = lo_all + n * i / threads, hi = lo_all + n * (i + 1) / threads;
            for (size_t j = lo; j < hi; j++) bits64[j] = stage[j];
        };
        std::vector<std::thread> th;
        for (unsigned i = 1; i < threads; i++) th.emplace_back(work, i);
        work(0);
        for (auto &x : th) x.join();
    };
    // Large batches: a three-stream pipeline over chunks of codewords.  The pageable upload is what bounds the call (1.08 GB
    // at 56 GB/s = 19 ms for the config-2 batch; cutting it into up to eight pieces costs nothing, scripts/micro/h2d_probe.py);
    // decode, download and widening of chunk c run behind the upload of chunk c+1 -- the upload on its own stream (a chunk's
    // kernels must not sit in front of the next chunk's copy), the download on a third (opposite PCIe direction), the
    // widening on a worker thread while this thread is blocked inside the next pageable copy.  Measured: 26.2 -> 22.4 ms
    // for the config-2 batch (3.0 G info-bit/s; 0.7 ms less with the last chunk on the state-per-lane kernels, not kept: one
    // kernel family per call keeps cpx_last_kernel and the precision mode unambiguous).  (Two earlier pipelines lost: chunks on two streams with the
    // kernels in the upload stream, 44.7 vs 39.2 ms, and host threads copying into pinned staging blocks, 42.6 vs 40.3 ms:
    // an extra pass over host memory costs more than it hides.)
    const int64_t min_chunk = 4096;
    int nch = (int)std::min<int64_t>(8, B / min_chunk);
    if (nch >= 2 && (size_t)(B * len) * sizeof(double) >= ((size_t)128 << 20)) {
        static hipStream_t s_up[64] = {}, s_cmp[64] = {}, s_dn[64] = {};     // per device, created once (under `mu`)
        int dev = 0;
        CPX_HIP(hipGetDevice(&dev));
        CPX_REQUIRE(dev >= 0 && dev < 64, CPX_ELIMIT, "viterbi: device index out of range");
        if (!s_up[dev]) {
            CPX_HIP(hipStreamCreateWithFlags(&s_up[dev], hipStreamNonBlocking));
            CPX_HIP(hipStreamCreateWithFlags(&s_cmp[dev], hipStreamNonBlocking));
            CPX_HIP(hipStreamCreateWithFlags(&s_dn[dev], hipStreamNonBlocking));
        }
        std::vector<hipEvent_t> ev_up(nch, nullptr), ev_cmp(nch, nullptr), ev_dn(nch, nullptr);
        bool ev_ok = true;
        for (int c = 0; c < nch && ev_ok; c++)
            ev_ok = hipEventCreateWithFlags(&ev_up[c], hipEventDisableTiming) == hipSuccess &&
                    hipEventCreateWithFlags(&ev_cmp[c], hipEventDisableTiming) == hipSuccess &&
                    hipEventCreateWithFlags(&ev_dn[c], hipEventDisableTiming) == hipSuccess;
        if (!ev_ok) {                                              // nothing is in flight yet: release what was created
            for (int c = 0; c < nch; c++)
                for (hipEvent_t e : {ev_up[c], ev_cmp[c], ev_dn[c]})
                    if (e) (void)hipEventDestroy(e);
            set_error("viterbi: hipEventCreate failed");
            return CPX_EHIP;
        }
        auto cw_lo = [&](int c) { return B * c / nch; };
        int issued = 0, wrc = CPX_OK;
        std::mutex qmu;
        std::condition_variable qcv;
        std::thread widener([&] {                                  // widens chunk c as soon as its download has landed
            (void)hipSetDevice(dev);                               // a new host thread starts on device 0
            for (int c = 0; c < nch; c++) {
                {
                    std::unique_lock<std::mutex> ql(qmu);
                    qcv.wait(ql, [&] { return issued > c || issued < 0; });
                    if (issued < 0) return;
                }
                if (hipEventSynchronize(ev_dn[c]) != hipSuccess) { wrc = CPX_EHIP; return; }
                CPX_TRACE("widen chunk (worker thread)");
                widen((size_t)(cw_lo(c) * L), (size_t)(cw_lo(c + 1) * L), nt > 2 ? nt - 1 : 1);
            }
        });
        auto fail = [&](int code) {
            { std::lock_guard<std::mutex> ql(qmu); issued = -1; }
            qcv.notify_all();
            widener.join();
            (void)hipDeviceSynchronize();
            for (int c = 0; c < nch; c++) { (void)hipEventDestroy(ev_up[c]); (void)hipEventDestroy(ev_cmp[c]); (void)hipEventDestroy(ev_dn[c]); }
            return code;
        };
        for (int c = 0; c < nch; c++) {
            const int64_t lo = cw_lo(c), n = cw_lo(c + 1) - lo;
            {
                CPX_TRACE("H2D chunk");
                if (hipMemcpyAsync(din.as<double>() + lo * len, coded + lo * len, sizeof(double) * (size_t)(n * len),
                                   hipMemcpyHostToDevice, s_up[dev]) != hipSuccess ||
                    hipEventRecord(ev_up[c], s_up[dev]) != hipSuccess) { set_error("viterbi: upload failed"); return fail(CPX_EHIP); }
            }
            if (hipStreamWaitEvent(s_cmp[dev], ev_up[c], 0) != hipSuccess) { set_error("viterbi: stream wait failed"); return fail(CPX_EHIP); }
            viterbi_prefer_cw(true);                               // see viterbi_cw.hip: the chunk's round hides behind the next upload
            rc = cpx_viterbi_decode_batch_dev(t, din.as<double>() + lo * len, n, len, L, n_steps, tb_depth, decoding_type,
                                              dout.as<uint8_t>() + lo * L, s_cmp[dev]);
            viterbi_prefer_cw(false);
            if (rc) return fail(rc);
            if (hipEventRecord(ev_cmp[c], s_cmp[dev]) != hipSuccess || hipStreamWaitEvent(s_dn[dev], ev_cmp[c], 0) != hipSuccess ||
                hipMemcpyAsync(stage + lo * L, dout.as<uint8_t>() + lo * L, (size_t)(n * L), hipMemcpyDeviceToHost, s_dn[dev]) != hipSuccess ||
                hipEventRecord(ev_dn[c], s_dn[dev]) != hipSuccess) { set_error("viterbi: download failed"); return fail(CPX_EHIP); }
            { std::lock_guard<std::mutex> ql(qmu); issued = c + 1; }
            qcv.notify_all();
        }
        widener.join();
        for (int c = 0; c < nch; c++) { (void)hipEventDestroy(ev_up[c]); (void)hipEventDestroy(ev_cmp[c]); (void)hipEventDestroy(ev_dn[c]); }
        if (wrc) { set_error("viterbi: download failed"); return wrc; }
        CPX_HIP(hipStreamSynchronize(s_cmp[dev]));
        return CPX_OK;
    }
    hipStream_t st = lib_stream();
    {
        CPX_TRACE("H2D coded");
        CPX_HIP(hipMemcpyAsync(din.p, coded, sizeof(double) * (size_t)(B * len), hipMemcpyHostToDevice, st));
        if (trace_enabled()) CPX_HIP(hipStreamSynchronize(st));   // ranges time the phases only if they do not overlap
    }
    rc = cpx_viterbi_decode_batch_dev(t, din.as<double>(), B, len, L, n_steps, tb_depth, decoding_type,
                                      dout.as<uint8_t>(), st);
    if (rc) return rc;
    {
        CPX_TRACE("D2H bits");
        CPX_HIP(hipMemcpyAsync(stage, dout.p, nout, hipMemcpyDeviceToHost, st));
        CPX_HIP(hipStreamSynchronize(st));
    }
    CPX_TRACE("widen to int64 (host threads)");
    widen(0, nout, nout < (1u << 20) ? 1 : nt);
    return CPX_OK;
}

}  // extern "C"
