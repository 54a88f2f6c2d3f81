// BCJR / MAP decoder and fused turbo decoder for gfx950.  Replaces the bodies of
//   map_decode    (/root/reference/commpy/channelcoding/turbo.py:163-251)
//     _backward_recursion (:78-111), _forward_recursion_decoding (:114-158), _compute_branch_prob (:62-76)
//   turbo_decode  (turbo.py:254-333) with interlv/deinterlv (interleavers.py:13-47)
// Probability-domain recursions, float64, the reference's formulas (no log-MAP):
//   gamma = exp(-((r0-c0)^2+(r1-c1)^2)/(2*nv)), priors p0 = 1/(1+e^L), p1 = 1-p0, beta_N = 1 (unterminated),
//   alpha_0 = delta(state 0), L = L_int + log(app1/app0) (app without prior).
//
// Mapping (wave64): lane = g*S + s -- GW <= 64/S codewords per wavefront, one trellis state per lane -- and TWO
// wavefronts per group of GW codewords: the alpha recursion (:114-158) and the beta recursion (:78-111) of a MAP pass are
// independent until the a-posteriori combine, so a FORWARD wave and a REVERSE wave run them concurrently from the two
// ends of the block and cross in the middle:
//   phase 1   F: alpha over the first half of the chunks,  R: beta over the second half; each keeps only its state
//             vector at the chunk boundaries (one 512-byte row per chunk in an HBM scratch -- a checkpoint, not a row
//             per step);
//   phase 2   F continues alpha over the second half: per chunk it RECOMPUTES the chunk's beta rows from R's checkpoint
//             (same operations, registers only), then runs alpha and the combine; R continues beta over the first half,
//             recomputing the chunk's alpha rows from F's checkpoint.
// (Round 1 ran beta over the whole block, streamed every beta row to HBM, then alpha: one wave per SIMD on a dependent
//  float64 chain -- rocprofv3: VALU busy 38 %, 41 % of the wave cycles waiting, 30 GB of HBM traffic per config-3 launch.
//  Now the chain a wave walks is half as long, two waves share a SIMD, and the HBM traffic of a pass is its inputs, its
//  LLRs and 1/8 row per step.)
// Time is processed in chunks of CH steps; every chunk has
//   (1) a TIME-PARALLEL stage: the 64 lanes load the chunk's received values (coalesced segments, prefetched one chunk
//       ahead), evaluate the four distinct branch probabilities and the two priors of every (codeword, step) item --
//       all the exp() work, off the serial chain -- and park them in LDS;
//   (2) the SERIAL recursion(s) over the chunk: alpha/beta of a state live in one VGPR pair of its lane; neighbours'
//       values are exchanged by DPP quad permutes (4-state trellises) or through a 512-byte LDS buffer; sums over the
//       states of a codeword are DPP butterflies (no LDS).  A full chunk is unrolled (compile-time LDS offsets);
//   (3) phase 2 only: the per-lane branch products alpha*gamma*beta are parked in LDS and a time-parallel epilogue adds
//       them in state order, divides, takes the log and writes the LLRs as coalesced segments.
// The sum-normalisation of the reference (turbo.py:110-111, :155-158) is applied every KNORM = 4 steps with the hardware
// reciprocal: a common positive factor per time step cancels in app1/app0 and in every later normalisation, so LLRs
// differ from the reference's only by rounding (measured <= 1e-13; tolerance 1e-5).
// Waves of a workgroup only meet at the phase boundary and between the stages of turbo_decode (__syncthreads +
// agent-scope fences: the partner's checkpoints / LLRs are read through L2).  turbo_decode runs its whole iteration loop
// inside ONE launch; the interleaver is a gather/scatter through per-codeword L arrays in a slab.
#include "cpx_internal.h"
#include "cpx_math.h"

#include <cstdlib>

using namespace cpx;

namespace {

constexpr int CH = 8;          // steps per chunk
constexpr int KNORM = 4;       // renormalise alpha / beta every KNORM steps
constexpr int NPAIR = 4;       // (forward, reverse) wave pairs per full-size workgroup: 8 waves = 2 per SIMD of a CU

struct MapTables {
    const int32_t *next_state, *output;                       // [S][2]
    const int32_t *pred_state, *pred_input, *pred_code;       // [S][2]
    int lgS, n, sr4;
};

template <int CTRL>
__device__ __forceinline__ double dppd(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// all-reduce sum inside aligned groups of 2^LGS lanes (LGS <= 4: DPP only)
template <int LGS>
__device__ __forceinline__ double group_sum(double v) {
    if (LGS >= 1) v += dppd<0xB1>(v);     // quad_perm [1,0,3,2]
    if (LGS >= 2) v += dppd<0x4E>(v);     // quad_perm [2,3,0,1]
    if (LGS >= 3) v += dppd<0x141>(v);    // row_half_mirror
    if (LGS >= 4) v += dppd<0x140>(v);    // row_mirror
    return v;
}

// One wavefront of a pair.  GW = codewords decoded by the pair (power of two, <= 64/S and <= 16): when the batch
// cannot fill the chip the host picks GW < 64/S (idle lanes, more pairs).
template <int LGS>
struct Ctx {
    static constexpr int S = 1 << LGS;
    static constexpr int NI = 2;                                       // (codeword, step) items per lane: CH*GW <= 128
    int lane, g, s, GW, P;                                             // P: doubles per step row of `tab`
    bool fwd;                                                          // forward (alpha) wave of the pair
    int sb[2];                                                         // sr4: MSB of next_state[s][i] (which successor input i leads to)
    bool active;                                                       // lane belongs to one of the GW decoded slots
    int nxt[2], code[2];              // outgoing branches of state s: next-state lane, 2-bit code (sys, parity)
    int plane[2], pin[2], pcode[2];   // incoming branches in np.where order: predecessor lane, input, code
    // LDS of this wave
    double *tab;    // [CH][GW][6] (+2 pad per step): gamma[4], p0, p1 of every (step, codeword) item
    double *xs;     // [CH][64][2]  per-lane branch products alpha*gamma*beta of the chunk (phase 2)
    double *xch;    // [64]         exchange buffer (trellises without the DPP fast path)
};

template <int LGS>
__host__ __device__ constexpr size_t wave_lds_doubles(int GW) { return (size_t)CH * (GW * 6 + 2) + (size_t)CH * 64 * 2 + 64; }

template <int LGS>
__device__ __forceinline__ void init_ctx(Ctx<LGS> &c, const MapTables &tb, unsigned char *smem, int GW) {
    constexpr int S = Ctx<LGS>::S;
    const int wave = threadIdx.x >> 6;
    c.lane = threadIdx.x & 63;
    c.fwd = (wave & 1) == 0;
    c.GW = GW;
    c.P = GW * 6 + 2;
    c.active = (c.lane >> LGS) < GW;
    c.g = c.active ? (c.lane >> LGS) : 0;                         // idle lanes shadow slot 0 (reads only)
    c.s = c.lane & (S - 1);
    const int gbase = (c.lane >> LGS) << LGS, sh = tb.n - 2;      // exchanges stay inside the lane's own group
    for (int i = 0; i < 2; i++) {
        c.nxt[i] = gbase + tb.next_state[c.s * 2 + i];
        c.sb[i] = (tb.next_state[c.s * 2 + i] >> (LGS - 1)) & 1;
        c.code[i] = (tb.output[c.s * 2 + i] >> sh) & 3;       // [msg_bit, parity_bit] = codeword_array[0:2] (:96-98)
        c.plane[i] = gbase + tb.pred_state[c.s * 2 + i];
        c.pin[i] = tb.pred_input[c.s * 2 + i];
        c.pcode[i] = (tb.pred_code[c.s * 2 + i] >> sh) & 3;
    }
    double *p = reinterpret_cast<double *>(smem) + (size_t)wave * wave_lds_doubles<LGS>(GW);
    c.tab = p; p += CH * c.P;
    c.xs = p;  p += CH * 64 * 2;
    c.xch = p;
}

// value of `v` in lane (quad base + idx), idx in 0..3 per lane: 4 quad broadcasts + selects (no LDS)
__device__ __forceinline__ double quad_fetch(double v, int idx) {
    const double q0 = dppd<0x00>(v), q1 = dppd<0x55>(v), q2 = dppd<0xAA>(v), q3 = dppd<0xFF>(v);
    const double lo = (idx & 1) ? q1 : q0, hi = (idx & 1) ? q3 : q2;
    return (idx & 2) ? hi : lo;
}

// neighbour exchange: values of `v` held by lanes la and lb of the same codeword
template <int LGS>
__device__ __forceinline__ void exchange2(const Ctx<LGS> &c, double v, int la, int lb, double &va, double &vb) {
    if (LGS == 2) {
        va = quad_fetch(v, la & 3);
        vb = quad_fetch(v, lb & 3);
    } else {
        c.xch[c.lane] = v;
        asm volatile("" ::: "memory");                            // in-order LDS: the reads below see this step's values
        va = c.xch[la];
        vb = c.xch[lb];
        asm volatile("" ::: "memory");
    }
}

// beta of the two successors of this lane's state, ordered by INPUT (0, 1).  4-state shift-register trellises:
// the successors of state s are (s>>1) and 2|(s>>1) -- two constant quad permutes + a per-lane select.
template <int LGS, bool SR>
__device__ __forceinline__ void exchange_succ(const Ctx<LGS> &c, double v, double &v0, double &v1) {
    if (LGS == 2 && SR) {
        const double lo = dppd<0x50>(v);                          // quad_perm [0,0,1,1]: lane s <- lane s>>1
        const double hi = dppd<0xFA>(v);                          // quad_perm [2,2,3,3]: lane s <- lane 2|(s>>1)
        v0 = c.sb[0] ? hi : lo;
        v1 = c.sb[1] ? hi : lo;
    } else {
        exchange2<LGS>(c, v, c.nxt[0], c.nxt[1], v0, v1);
    }
}

// alpha of the two predecessors of this lane's state in np.where order (increasing predecessor state).
// 4-state shift-register trellises: predecessors of ns are 2(ns&1) and 2(ns&1)+1.
template <int LGS, bool SR>
__device__ __forceinline__ void exchange_pred(const Ctx<LGS> &c, double v, double &v0, double &v1) {
    if (LGS == 2 && SR) {
        v0 = dppd<0x88>(v);                                       // quad_perm [0,2,0,2]
        v1 = dppd<0xDD>(v);                                       // quad_perm [1,3,1,3]
    } else {
        exchange2<LGS>(c, v, c.plane[0], c.plane[1], v0, v1);
    }
}

// ---- time-parallel stage of a chunk --------------------------------------------------------------------------
// A chunk has CH*GW (codeword, step) items, at most 2 per lane: item p = lane + 64*q -> codeword slot p / CH,
// step p % CH (consecutive lanes -> consecutive steps of one codeword: coalesced 64-byte segments).
struct RawChunk {
    double r0[2], r1[2], li[2];
};

struct PassIO {
    const double *sys;            // [B][N]
    const int32_t *sys_perm;      // null, or sys is read through the interleaver (sys_symbols_i = interlv(sys), turbo.py:310)
    const double *par;            // [B][N]
    const double *Lin;            // L_int, stride lstride per codeword
    double *Lout;                 // L_int + log(app1/app0), stride lstride
    int64_t lstride, cw0, B, N;
    double nv2;
    double *ckpt;                 // this pair's checkpoint rows [nchunks + 1][64]
};

template <int LGS>
__device__ __forceinline__ void load_raw(const Ctx<LGS> &c, const PassIO &io, RawChunk &rc, int64_t t0, int len) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int p = c.lane + 64 * q, gg = p / CH, tl = p % CH;
        const int64_t cw = io.cw0 + gg, t = t0 + tl;              // 0-based step index
        rc.r0[q] = 0.0; rc.r1[q] = 0.0; rc.li[q] = 0.0;
        if (gg < c.GW && cw < io.B && tl < len) {
            rc.r0[q] = io.sys[cw * io.N + (io.sys_perm ? io.sys_perm[t] : t)];
            rc.r1[q] = io.par[cw * io.N + t];
            rc.li[q] = io.Lin[cw * io.lstride + t];
        }
    }
}

// tab[tl][g] = gamma[0..3] (_compute_branch_prob :62-76), p0, p1 (priors :239-240) into LDS.  The row stride P = 6 GW + 2
// doubles makes the eight lanes that hold consecutive steps of a codeword hit eight different 16-byte bank groups.
template <int LGS>
__device__ __forceinline__ void stage_chunk(const Ctx<LGS> &c, const RawChunk &rc, double nv2) {
    const int GW = c.GW;
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int p = c.lane + 64 * q, gg = p / CH, tl = p % CH;
        if (gg < GW) {
            double g4[4];
#pragma unroll
            for (int code = 0; code < 4; code++) {
                const double x = rc.r0[q] - (double)(2 * ((code >> 1) & 1) - 1);
                const double y = rc.r1[q] - (double)(2 * (code & 1) - 1);
                g4[code] = exp(-(x * x + y * y) / nv2);
            }
            const double p0 = 1.0 / (1.0 + exp(rc.li[q]));
            double2 *row = reinterpret_cast<double2 *>(c.tab + tl * c.P + gg * 6);
            row[0] = make_double2(g4[0], g4[1]);
            row[1] = make_double2(g4[2], g4[3]);
            row[2] = make_double2(p0, 1.0 - p0);                  // priors[1] = 1 - priors[0] (:240)
        }
    }
    asm volatile("" ::: "memory");
}

// One beta step (:106-111): b <- sum_i b[next(s,i)] * gamma(code(s,i)) * prior(i), inputs 0 then 1.
// With X = true also parks x_i = a_own * gamma(code(s,i)) * b[next(s,i)] (the branch products of :141-143).
template <int LGS, bool SR, bool X>
__device__ __forceinline__ void beta_step(const Ctx<LGS> &c, int tl, double &b, double a_own) {
    const double *it = c.tab + tl * c.P + c.g * 6;
    const double g0 = it[c.code[0]], g1 = it[c.code[1]], p0 = it[4], p1 = it[5];
    double bn0, bn1;
    exchange_succ<LGS, SR>(c, b, bn0, bn1);
    if (X) {
        double2 xv;
        xv.x = a_own * g0 * bn0;
        xv.y = a_own * g1 * bn1;
        *reinterpret_cast<double2 *>(c.xs + (tl * 64 + c.lane) * 2) = xv;   // idle lanes own a slot too
    }
    double nb = 0.0;
    nb += (bn0 * g0 * p0);
    nb += (bn1 * g1 * p1);
    b = nb;
    // (:110-111) every KNORM steps; any positive common factor is a valid normalisation, so the hardware
    // reciprocal (v_rcp_f64) is used instead of a correctly rounded division
    if ((tl & (KNORM - 1)) == 0) b = b * __builtin_amdgcn_rcp(group_sum<LGS>(b));
}

// One alpha step (:136-138, :155-158): a <- sum over the incoming branches in (predecessor state, input) order.
// With X = true first parks x_i = a * gamma(code(s,i)) * beta_next[next(s,i)] for this lane's outgoing branches.
template <int LGS, bool SR, bool X>
__device__ __forceinline__ void alpha_step(const Ctx<LGS> &c, int tl, double &a, double beta_row) {
    const double *it = c.tab + tl * c.P + c.g * 6;
    const double gi0 = it[c.pcode[0]], gi1 = it[c.pcode[1]];
    const double q0 = it[4 + c.pin[0]], q1 = it[4 + c.pin[1]];
    if (X) {
        const double go0 = it[c.code[0]], go1 = it[c.code[1]];
        double bt0, bt1;
        exchange_succ<LGS, SR>(c, beta_row, bt0, bt1);
        double2 xv;
        xv.x = a * go0 * bt0;
        xv.y = a * go1 * bt1;
        *reinterpret_cast<double2 *>(c.xs + (tl * 64 + c.lane) * 2) = xv;   // idle lanes own a slot too
    }
    double ap0, ap1;
    exchange_pred<LGS, SR>(c, a, ap0, ap1);
    double na = 0.0;
    na += (ap0 * gi0 * q0);
    na += (ap1 * gi1 * q1);
    a = na;
    if ((tl & (KNORM - 1)) == KNORM - 1) a = a * __builtin_amdgcn_rcp(group_sum<LGS>(a));
}

// time-parallel epilogue of a phase-2 chunk: app sums in state order, L = L_int + log(app1/app0) (:145)
template <int LGS>
__device__ __forceinline__ void epilogue(const Ctx<LGS> &c, const PassIO &io, const RawChunk &rc, int64_t t_lo, int len) {
    constexpr int S = Ctx<LGS>::S;
    asm volatile("" ::: "memory");
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int p = c.lane + 64 * q, gg = p / CH, tl = p % CH;
        const int64_t cw = io.cw0 + gg;
        if (gg < c.GW && cw < io.B && tl < len) {
            const double2 *x = reinterpret_cast<const double2 *>(c.xs + (tl * 64 + gg * S) * 2);
            double app0 = 0.0, app1 = 0.0;
#pragma unroll
            for (int st = 0; st < S; st++) { const double2 v = x[st]; app0 += v.x; app1 += v.y; }
            io.Lout[cw * io.lstride + t_lo + tl] = rc.li[q] + fast_log(app1 / app0);
        }
    }
    asm volatile("" ::: "memory");
}

// The two waves of a pair belong to one workgroup, i.e. one CU and one (write-through) vector L1: workgroup-scope
// fences order their global stores and loads.  (Agent-scope fences -- __threadfence() -- write back / invalidate the
// XCD's whole L2 on gfx950, whose L2s are not coherent with each other: measured 15.8 ms instead of 12.3 ms per
// config-3 launch with eight of them per iteration.)
__device__ __forceinline__ void pair_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// One MAP pass over the GW codewords of this pair of wavefronts.  Collective over the workgroup (one pair_sync).
template <int LGS, bool SR>
__device__ void map_pass(const Ctx<LGS> &c, const PassIO &io) {
    const int64_t N = io.N;
    const int K = (int)((N + CH - 1) / CH), K1 = K / 2;          // F: chunks [0,K1) then [K1,K); R: [K1,K) then [0,K1)
    auto clen = [&](int k) { const int64_t r = N - (int64_t)k * CH; return (int)(r < CH ? r : CH); };
    double *ck = io.ckpt + c.lane;                                // row k: state vector at time k*CH
    RawChunk cur, nxt;
    if (c.fwd) {
        // ---------------- phase 1: alpha over chunks 0 .. K1-1, checkpoint before every chunk ----------------
        double a = (c.s == 0) ? 1.0 : 0.0;                        // f_state_metrics[0][0] = 1 (:221)
        if (K1 > 0) load_raw<LGS>(c, io, cur, 0, clen(0));
        for (int k = 0; k < K1; k++) {
            ck[(int64_t)k * 64] = a;                              // alpha at time k*CH (read by R in phase 2)
            stage_chunk<LGS>(c, cur, io.nv2);
            if (k + 1 < K1) load_raw<LGS>(c, io, nxt, (int64_t)(k + 1) * CH, clen(k + 1));
            const int len = clen(k);
            if (len == CH) {
#pragma unroll
                for (int tl = 0; tl < CH; tl++) alpha_step<LGS, SR, false>(c, tl, a, 0.0);
            } else {
                for (int tl = 0; tl < len; tl++) alpha_step<LGS, SR, false>(c, tl, a, 0.0);
            }
            cur = nxt;
        }
        load_raw<LGS>(c, io, cur, (int64_t)K1 * CH, clen(K1));    // first chunk of phase 2 (K1 < K always)
        pair_sync();
        // ---------------- phase 2: chunks K1 .. K-1: beta rows from R's checkpoint, alpha + combine ----------------
        double bup = ck[(int64_t)(K1 + 1) * 64];                  // beta at the upper boundary of chunk K1
        for (int k = K1; k < K; k++) {
            const int len = clen(k);
            stage_chunk<LGS>(c, cur, io.nv2);
            double bupn = 0.0;
            if (k + 1 < K) {
                load_raw<LGS>(c, io, nxt, (int64_t)(k + 1) * CH, clen(k + 1));
                bupn = ck[(int64_t)(k + 2) * 64];
            }
            double brow[CH];                                      // brow[tl] = beta at time k*CH + tl + 1
            double b = bup;
            if (len == CH) {
#pragma unroll
                for (int tl = CH - 1; tl >= 0; --tl) { brow[tl] = b; beta_step<LGS, SR, false>(c, tl, b, 0.0); }
#pragma unroll
                for (int tl = 0; tl < CH; tl++) alpha_step<LGS, SR, true>(c, tl, a, brow[tl]);
            } else {
#pragma unroll
                for (int tl = CH - 1; tl >= 0; --tl)
                    if (tl < len) { brow[tl] = b; beta_step<LGS, SR, false>(c, tl, b, 0.0); }
#pragma unroll
                for (int tl = 0; tl < CH; tl++)
                    if (tl < len) alpha_step<LGS, SR, true>(c, tl, a, brow[tl]);
            }
            epilogue<LGS>(c, io, cur, (int64_t)k * CH, len);
            cur = nxt;
            bup = bupn;
        }
    } else {
        // ---------------- phase 1: beta over chunks K-1 .. K1, checkpoint before every chunk ----------------
        double b = 1.0;                                           // b_state_metrics[:, N] = 1 (:225)
        load_raw<LGS>(c, io, cur, (int64_t)(K - 1) * CH, clen(K - 1));
        for (int k = K - 1; k >= K1; --k) {
            ck[(int64_t)(k + 1) * 64] = b;                        // beta at the upper boundary of chunk k (read by F)
            stage_chunk<LGS>(c, cur, io.nv2);
            if (k > K1) load_raw<LGS>(c, io, nxt, (int64_t)(k - 1) * CH, CH);
            const int len = clen(k);
            if (len == CH) {
#pragma unroll
                for (int tl = CH - 1; tl >= 0; --tl) beta_step<LGS, SR, false>(c, tl, b, 0.0);
            } else {
                for (int tl = len - 1; tl >= 0; --tl) beta_step<LGS, SR, false>(c, tl, b, 0.0);
            }
            cur = nxt;
        }
        if (K1 > 0) load_raw<LGS>(c, io, cur, (int64_t)(K1 - 1) * CH, CH);
        pair_sync();
        // ---------------- phase 2: chunks K1-1 .. 0: alpha rows from F's checkpoint, beta + combine ----------------
        double alo = K1 > 0 ? ck[(int64_t)(K1 - 1) * 64] : 0.0;   // alpha at the lower boundary of chunk K1-1
        for (int k = K1 - 1; k >= 0; --k) {
            stage_chunk<LGS>(c, cur, io.nv2);
            double alon = 0.0;
            if (k > 0) {
                load_raw<LGS>(c, io, nxt, (int64_t)(k - 1) * CH, CH);
                alon = ck[(int64_t)(k - 1) * 64];
            }
            double arow[CH];                                      // arow[tl] = alpha at time k*CH + tl (full chunks only here)
            double a = alo;
#pragma unroll
            for (int tl = 0; tl < CH; tl++) { arow[tl] = a; alpha_step<LGS, SR, false>(c, tl, a, 0.0); }
#pragma unroll
            for (int tl = CH - 1; tl >= 0; --tl) beta_step<LGS, SR, true>(c, tl, b, arow[tl]);
            epilogue<LGS>(c, io, cur, (int64_t)k * CH, CH);
            cur = nxt;
            alo = alon;
        }
    }
}

struct MapParams {
    MapTables tb;
    const double *sys, *par, *Lin;     // [B][N]
    double *Lout;                      // [B][N]
    uint8_t *bits;                     // [B][N]
    double *scratch;                   // per pair: checkpoint rows [nchunks + 1][64]
    int64_t B, N;
    double nv2;
    int want_bits, GW;
};

template <int LGS, bool SR>
__global__ __launch_bounds__(128 * NPAIR) void map_decode_kernel(MapParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Ctx<LGS> c;
    init_ctx<LGS>(c, p.tb, smem, p.GW);
    const int64_t pair = (int64_t)blockIdx.x * (blockDim.x >> 7) + (threadIdx.x >> 7);
    const int64_t K = (p.N + CH - 1) / CH;
    PassIO io;
    io.sys = p.sys; io.sys_perm = nullptr; io.par = p.par; io.Lin = p.Lin; io.Lout = p.Lout; io.lstride = p.N;
    io.cw0 = pair * p.GW; io.B = p.B; io.N = p.N; io.nv2 = p.nv2;
    io.ckpt = p.scratch + pair * (K + 1) * 64;
    map_pass<LGS, SR>(c, io);
    pair_sync();
    // decoded_bits: L > 0 in 'decode' mode only (:148-152); the two waves of the pair take alternate codewords
    for (int g = (threadIdx.x >> 6) & 1; g < p.GW; g += 2) {
        const int64_t cw = io.cw0 + g;
        if (cw >= p.B) break;
        for (int64_t t = c.lane; t < p.N; t += 64)
            p.bits[cw * p.N + t] = (uint8_t)((p.want_bits && p.Lout[cw * p.N + t] > 0) ? 1 : 0);
    }
}

struct TurboParams {
    MapTables tb;
    const double *sys, *p1, *p2, *Lint;   // [B][N], Lint may be null
    const int32_t *perm;                  // [N]
    uint8_t *bits;                        // [B][N]
    double *ckpt;                         // per pair: checkpoint rows [nchunks + 1][64]
    double *larr;                         // per codeword: A[N] B[N] C[N]
    int64_t B, N;
    double nv2;
    int n_iter, GW;
};

template <int LGS, bool SR>
__global__ __launch_bounds__(128 * NPAIR) void turbo_decode_kernel(TurboParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Ctx<LGS> c;
    init_ctx<LGS>(c, p.tb, smem, p.GW);
    const int64_t pair = (int64_t)blockIdx.x * (blockDim.x >> 7) + (threadIdx.x >> 7);
    const int64_t N = p.N, K = (N + CH - 1) / CH, cw0 = pair * p.GW;
    // L arrays of all codewords: [B][3][N] -> per-codeword stride 3N; A at +0, B at +N, C at +2N
    double *A0 = p.larr, *B0 = p.larr + N, *C0 = p.larr + 2 * N;
    const int64_t ls = 3 * N;
    PassIO io;
    io.sys = p.sys; io.lstride = ls; io.cw0 = cw0; io.B = p.B; io.N = N; io.nv2 = p.nv2;
    io.ckpt = p.ckpt + pair * (K + 1) * 64;
    // The element-wise stages between the MAP passes are done wave-wide, one codeword after the other (the two waves of
    // the pair take alternate codewords), 64 consecutive elements per instruction and four independent rounds in flight.
    const int lane = c.lane, GW = p.GW, w2 = (threadIdx.x >> 6) & 1;
    for (int g = w2; g < GW; g += 2) {
        const int64_t cwg = cw0 + g;
        if (cwg >= p.B) break;
        double *A = A0 + cwg * ls;
#pragma unroll 4
        for (int64_t t = lane; t < N; t += 64) A[t] = p.Lint ? p.Lint[cwg * N + t] : 0.0;      // L_int_1 (:305-308)
    }
    pair_sync();
    // 2 n_iter half-iterations through ONE map_pass call site (two inlined copies of the unrolled pass spill registers):
    //   even h: [L_ext_1, _] = map_decode(sys, non_sys_1, trellis, nv, L_int_1, 'compute')          (:315)
    //           L_ext_1 -= L_int_1 ; L_int_2 = interlv(L_ext_1)                                     (:318-319)
    //   odd h:  [L_2, bits] = map_decode(sys_i, non_sys_2, trellis, nv, L_int_2, mode)              (:326)
    //           L_ext_2 = L_2 - L_int_2 ; L_int_1 = deinterlv(L_ext_2)                              (:328-329)
    for (int h = 0; h < 2 * p.n_iter; h++) {
        const bool second = h & 1;
        io.sys_perm = second ? p.perm : nullptr;
        io.par = second ? p.p2 : p.p1;
        io.Lin = second ? C0 : A0;
        io.Lout = B0;
        map_pass<LGS, SR>(c, io);
        pair_sync();
        for (int g = w2; g < GW; g += 2) {
            const int64_t cwg = cw0 + g;
            if (cwg >= p.B) break;
            double *A = A0 + cwg * ls, *C = C0 + cwg * ls;
            const double *Bb = B0 + cwg * ls;
            if (!second) {
#pragma unroll 4
                for (int64_t t = lane; t < N; t += 64) {      // same subtraction, fused with the interleaver gather
                    const int32_t q = p.perm[t];
                    C[t] = Bb[q] - A[q];
                }
            } else {
#pragma unroll 4
                for (int64_t t = lane; t < N; t += 64) A[p.perm[t]] = Bb[t] - C[t];
            }
        }
        pair_sync();
    }
    // decoded_bits = deinterlv(decoded_bits of the last MAP2)                              (:331)
    for (int g = w2; g < GW; g += 2) {
        const int64_t cwg = cw0 + g;
        if (cwg >= p.B) break;
        const double *Bb = B0 + cwg * ls;
#pragma unroll 4
        for (int64_t t = lane; t < N; t += 64)
            p.bits[cwg * N + p.perm[t]] = (uint8_t)((p.n_iter > 0 && Bb[t] > 0) ? 1 : 0);
    }
}

// codewords per pair of wavefronts: full wavefronts as soon as the batch gives every SIMD of the chip its two waves; for
// smaller batches fewer codewords per pair (idle lanes) spread the work over more SIMDs.
int pick_gw(int S, int64_t B) {
    int G = 64 / S;
    if (G > 16) G = 16;                                            // CH*GW <= 128 items per chunk
    int gw = G;
    while (gw > 1 && (B + gw - 1) / gw < 1024) gw >>= 1;
    return gw;
}

// pairs per workgroup: four (eight waves, one workgroup per CU at full size) when there are enough pairs to give every
// CU such a workgroup, else one pair per workgroup
int pick_npair(int64_t npairs) {
    static const int forced = [] { const char *e = getenv("CPX_BCJR_NPAIR"); return e ? atoi(e) : 0; }();   // experiments
    if (forced == 1 || forced == 2 || forced == NPAIR) return forced;
    return npairs >= (int64_t)NPAIR * device_cus() ? NPAIR : 1;
}

int fill_tables(const cpx_trellis *t, MapTables &tb) {
    CPX_REQUIRE(t, CPX_EINVAL, "map_decode: null trellis");
    if (int rcd = check_handle_device(t->device, "map_decode")) return rcd;
    CPX_REQUIRE(t->I == 2 && t->k == 1, CPX_ELIMIT, "map_decode: only k = 1 (two inputs per step) trellises are supported, like the reference's priors[2]");
    CPX_REQUIRE(t->n >= 2, CPX_EINVAL, "map_decode: needs a rate-1/2 trellis (n >= 2)");
    CPX_REQUIRE(t->S >= 2 && t->S <= 16, CPX_ELIMIT, "map_decode: 2..16 states supported (got %d)", t->S);
    tb.next_state = t->d_next; tb.output = t->d_out;
    tb.pred_state = t->d_pred_state; tb.pred_input = t->d_pred_input; tb.pred_code = t->d_pred_code;
    tb.n = t->n;
    tb.lgS = 0;
    while ((1 << tb.lgS) < t->S) tb.lgS++;
    // 4-state shift-register structure (everything commpy's Trellis builds with memory 2, k = 1)
    tb.sr4 = (t->S == 4);
    for (int s2 = 0; s2 < 4 && tb.sr4; s2++) {
        const int a = t->next_state[s2 * 2], b2 = t->next_state[s2 * 2 + 1];
        if (!((a == (s2 >> 1) && b2 == (2 | (s2 >> 1))) || (b2 == (s2 >> 1) && a == (2 | (s2 >> 1))))) tb.sr4 = 0;
        if (t->pred_state[s2 * 2] != 2 * (s2 & 1) || t->pred_state[s2 * 2 + 1] != 2 * (s2 & 1) + 1) tb.sr4 = 0;
    }
    return CPX_OK;
}

}  // namespace

extern "C" {

int cpx_map_decode_batch_dev(const cpx_trellis *t, const double *d_sys, const double *d_par, const double *d_L_int,
                             int64_t B, int64_t N, double noise_variance, int want_bits, double *d_L_ext,
                             uint8_t *d_bits, void *stream) {
    MapParams p;
    int rc = fill_tables(t, p.tb);
    if (rc) return rc;
    CPX_REQUIRE(B >= 0 && N >= 0, CPX_EINVAL, "map_decode: negative size");
    if (B == 0 || N == 0) return CPX_OK;
    hipStream_t st = pick_stream(stream);
    const int GW = pick_gw(t->S, B);
    const int64_t npairs = (B + GW - 1) / GW;
    const int np = pick_npair(npairs);
    const int64_t nblocks = (npairs + np - 1) / np, K = (N + CH - 1) / CH;
    p.GW = GW;
    p.sys = d_sys; p.par = d_par; p.Lin = d_L_int; p.Lout = d_L_ext; p.bits = d_bits;
    p.B = B; p.N = N; p.nv2 = 2 * noise_variance; p.want_bits = want_bits;
    CPX_REQUIRE(nblocks < (1ll << 31), CPX_ELIMIT, "map_decode: batch too large");
    if ((rc = workspace(st, 0, sizeof(double) * (size_t)(nblocks * np * (K + 1) * 64), (void **)&p.scratch))) return rc;
    dim3 grid((unsigned)nblocks), block(128 * np);
    switch (p.tb.lgS) {
#define CASE(LG) case LG: hipLaunchKernelGGL((map_decode_kernel<LG, false>), grid, block, sizeof(double) * 2 * np * wave_lds_doubles<LG>(GW), st, p); break;
        case 2:                                                   // 4 states: the shift-register fast path where it applies
            if (p.tb.sr4) hipLaunchKernelGGL((map_decode_kernel<2, true>), grid, block, sizeof(double) * 2 * np * wave_lds_doubles<2>(GW), st, p);
            else hipLaunchKernelGGL((map_decode_kernel<2, false>), grid, block, sizeof(double) * 2 * np * wave_lds_doubles<2>(GW), st, p);
            break;
        CASE(1) CASE(3) CASE(4)
#undef CASE
        default: set_error("map_decode: unsupported state count"); return CPX_ELIMIT;
    }
    CPX_HIP(hipGetLastError());
    note_kernel("map_decode_kernel<%d,%s> (%d wave pairs per workgroup, %d codewords per pair)", p.tb.lgS, (p.tb.lgS == 2 && p.tb.sr4) ? "true" : "false", np, GW);
    return CPX_OK;
}

int cpx_turbo_decode_batch_dev(const cpx_trellis *t, const double *d_sys, const double *d_p1, const double *d_p2,
                               const double *d_L_int_or_null, const int32_t *d_perm, int64_t B, int64_t N,
                               double noise_variance, int n_iter, uint8_t *d_bits, void *stream) {
    TurboParams p;
    int rc = fill_tables(t, p.tb);
    if (rc) return rc;
    CPX_REQUIRE(B >= 0 && N >= 0 && n_iter >= 0, CPX_EINVAL, "turbo_decode: negative size");
    if (B == 0 || N == 0) return CPX_OK;
    hipStream_t st = pick_stream(stream);
    const int GW = pick_gw(t->S, B);
    const int64_t npairs = (B + GW - 1) / GW;
    const int np = pick_npair(npairs);
    const int64_t nblocks = (npairs + np - 1) / np, K = (N + CH - 1) / CH;
    p.GW = GW;
    p.sys = d_sys; p.p1 = d_p1; p.p2 = d_p2; p.Lint = d_L_int_or_null; p.perm = d_perm; p.bits = d_bits;
    p.B = B; p.N = N; p.nv2 = 2 * noise_variance; p.n_iter = n_iter;
    CPX_REQUIRE(nblocks < (1ll << 31), CPX_ELIMIT, "turbo_decode: batch too large");
    if ((rc = workspace(st, 0, sizeof(double) * (size_t)(nblocks * np * (K + 1) * 64), (void **)&p.ckpt))) return rc;
    if ((rc = workspace(st, 1, sizeof(double) * (size_t)(B * 3 * N), (void **)&p.larr))) return rc;
    dim3 grid((unsigned)nblocks), block(128 * np);
    switch (p.tb.lgS) {
#define CASE(LG) case LG: hipLaunchKernelGGL((turbo_decode_kernel<LG, false>), grid, block, sizeof(double) * 2 * np * wave_lds_doubles<LG>(GW), st, p); break;
        case 2:
            if (p.tb.sr4) hipLaunchKernelGGL((turbo_decode_kernel<2, true>), grid, block, sizeof(double) * 2 * np * wave_lds_doubles<2>(GW), st, p);
            else hipLaunchKernelGGL((turbo_decode_kernel<2, false>), grid, block, sizeof(double) * 2 * np * wave_lds_doubles<2>(GW), st, p);
            break;
        CASE(1) CASE(3) CASE(4)
#undef CASE
        default: set_error("turbo_decode: unsupported state count"); return CPX_ELIMIT;
    }
    CPX_HIP(hipGetLastError());
    note_kernel("turbo_decode_kernel<%d,%s> (%d wave pairs per workgroup, %d codewords per pair)", p.tb.lgS, (p.tb.lgS == 2 && p.tb.sr4) ? "true" : "false", np, GW);
    return CPX_OK;
}

int cpx_map_decode_batch(const cpx_trellis *t, const double *sys, const double *par, const double *L_int, int64_t B,
                         int64_t N, double noise_variance, int want_bits, double *L_ext, uint8_t *bits) {
    CPX_REQUIRE(t && ((sys && par && L_int && L_ext && bits) || B * N == 0), CPX_EINVAL, "map_decode: null pointer");
    int rc = ensure_device();
    if (rc) return rc;
    if (B == 0 || N == 0) return CPX_OK;
    const size_t nb = sizeof(double) * (size_t)(B * N);
    DevBuf ds, dp, dl, dout, dbits;
    if ((rc = ds.alloc(nb)) || (rc = dp.alloc(nb)) || (rc = dl.alloc(nb)) || (rc = dout.alloc(nb)) ||
        (rc = dbits.alloc((size_t)(B * N))))
        return rc;
    hipStream_t st = lib_stream();
    CPX_HIP(hipMemcpyAsync(ds.p, sys, nb, hipMemcpyHostToDevice, st));
    CPX_HIP(hipMemcpyAsync(dp.p, par, nb, hipMemcpyHostToDevice, st));
    CPX_HIP(hipMemcpyAsync(dl.p, L_int, nb, hipMemcpyHostToDevice, st));
    rc = cpx_map_decode_batch_dev(t, ds.as<double>(), dp.as<double>(), dl.as<double>(), B, N, noise_variance, want_bits,
                                  dout.as<double>(), dbits.as<uint8_t>(), st);
    if (rc) return rc;
    CPX_HIP(hipMemcpyAsync(L_ext, dout.p, nb, hipMemcpyDeviceToHost, st));
    CPX_HIP(hipMemcpyAsync(bits, dbits.p, (size_t)(B * N), hipMemcpyDeviceToHost, st));
    CPX_HIP(hipStreamSynchronize(st));
    return CPX_OK;
}

int cpx_turbo_decode_batch(const cpx_trellis *t, const double *sys, const double *p1, const double *p2,
                           const double *L_int_or_null, const int32_t *perm, int64_t B, int64_t N,
                           double noise_variance, int n_iter, uint8_t *bits) {
    CPX_REQUIRE(t && ((sys && p1 && p2 && perm && bits) || B * N == 0), CPX_EINVAL, "turbo_decode: null pointer");
    int rc = ensure_device();
    if (rc) return rc;
    if (B == 0 || N == 0) return CPX_OK;
    for (int64_t i = 0; i < N; i++)
        CPX_REQUIRE(perm[i] >= 0 && perm[i] < N, CPX_EINVAL, "turbo_decode: interleaver index out of range");
    const size_t nb = sizeof(double) * (size_t)(B * N);
    DevBuf ds, d1, d2, dl, dperm, dbits;
    if ((rc = ds.alloc(nb)) || (rc = d1.alloc(nb)) || (rc = d2.alloc(nb)) || (rc = dperm.alloc(sizeof(int32_t) * N)) ||
        (rc = dbits.alloc((size_t)(B * N))))
        return rc;
    if (L_int_or_null && (rc = dl.alloc(nb))) return rc;
    hipStream_t st = lib_stream();
    CPX_HIP(hipMemcpyAsync(ds.p, sys, nb, hipMemcpyHostToDevice, st));
    CPX_HIP(hipMemcpyAsync(d1.p, p1, nb, hipMemcpyHostToDevice, st));
    CPX_HIP(hipMemcpyAsync(d2.p, p2, nb, hipMemcpyHostToDevice, st));
    CPX_HIP(hipMemcpyAsync(dperm.p, perm, sizeof(int32_t) * N, hipMemcpyHostToDevice, st));
    if (L_int_or_null) CPX_HIP(hipMemcpyAsync(dl.p, L_int_or_null, nb, hipMemcpyHostToDevice, st));
    rc = cpx_turbo_decode_batch_dev(t, ds.as<double>(), d1.as<double>(), d2.as<double>(),
                                    L_int_or_null ? dl.as<double>() : nullptr, dperm.as<int32_t>(), B, N, noise_variance,
                                    n_iter, dbits.as<uint8_t>(), st);
    if (rc) return rc;
    CPX_HIP(hipMemcpyAsync(bits, dbits.p, (size_t)(B * N), hipMemcpyDeviceToHost, st));
    CPX_HIP(hipStreamSynchronize(st));
    return CPX_OK;
}

}  // extern "C"
