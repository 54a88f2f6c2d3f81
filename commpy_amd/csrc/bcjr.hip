// BCJR / MAP decoder and fused turbo decoder for gfx950.  Replaces the bodies of
//   map_decode    (/root/reference/commpy/channelcoding/turbo.py:163-251)
//     _backward_recursion (:78-111), _forward_recursion_decoding (:114-158), _compute_branch_prob (:62-76)
//   turbo_decode  (turbo.py:254-333) with interlv/deinterlv (interleavers.py:13-47)
// Probability-domain recursions, float64, the reference's formulas (no log-MAP):
//   gamma = exp(-((r0-c0)^2+(r1-c1)^2)/(2*nv)), priors p0 = 1/(1+e^L), p1 = 1-p0, beta_N = 1 (unterminated),
//   alpha_0 = delta(state 0), L = L_int + log(app1/app0) (app without prior).
//
// Mapping (wave64): lane = g*S + s -- G = 64/S codewords per wavefront, one trellis state per lane.
// Time is processed in chunks of CH steps; every chunk has
//   (1) a TIME-PARALLEL stage: the 64 lanes load the chunk's received values (coalesced segments), evaluate
//       the four distinct branch probabilities and the prior of every (codeword, step) pair -- all the exp()
//       work, off the serial chain -- and park them in LDS;
//   (2) the SERIAL recursion over the chunk: alpha/beta of a state live in one VGPR pair of its lane, the
//       neighbours' values are exchanged through a 512-byte LDS buffer (cheaper than ds_bpermute for
//       float64, see viterbi.hip), sums over the states of a codeword are DPP butterflies (no LDS).
// With one wavefront per SIMD (B = 16384 codewords fill the chip exactly once) the recursion is bound by the
// LATENCY of its dependent chain, so everything that is not alpha/beta itself is moved off it:
//   * the sum-normalisation of the reference (turbo.py:110-111, :155-158) is applied every KNORM = 4 steps
//     instead of every step (a common positive factor per time step cancels in app1/app0 and in every later
//     normalisation: LLRs differ from the reference's only by rounding, measured <= 1e-13; tolerance 1e-5);
//   * for 4-state trellises the neighbour exchange is 4 quad-broadcast DPP moves + per-lane selects (depth ~25
//     cycles) instead of an LDS round trip (~130 cycles);
//   * the a-posteriori sums are not reduced in the loop: every lane parks its two branch products in LDS and the
//     time-parallel epilogue of the chunk adds them, divides and takes the log;
//   * received values of the NEXT chunk are prefetched into registers before the serial loop of the current one.
// beta is CHECKPOINTED, not stored: the backward pass keeps beta only at chunk boundaries ([chunk][lane] rows in
// an HBM slab, 1/16 of the per-step traffic that made the first version HBM-bound); the forward pass recomputes
// the chunk's beta rows into LDS from the checkpoint with the same operations in the same order (bit-identical),
// then runs alpha over the chunk.
// Butterfly / epilogue sums differ from the reference's sequential order by O(1e-16) relative.
// LLRs of a chunk are staged in LDS, log() evaluated time-parallel, and written back as coalesced segments.  turbo_decode runs its whole iteration loop inside ONE launch; the
// interleaver is a gather/scatter through per-codeword L arrays in the slab.
#include "cpx_internal.h"
#include "cpx_math.h"

using namespace cpx;

namespace {

constexpr int MAXCH = 16;      // steps per chunk (upper bound; keeps LDS <= 37 KiB per wave)

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

// GW = codewords actually decoded by a wavefront (power of two, <= 64/S and <= 16).  When the batch cannot fill
// the chip with full wavefronts (B*S/64 < ~4 waves per SIMD) the host picks GW < 64/S: lanes of the unused
// codeword slots idle, but four times as many wavefronts hide the latency of the serial recursion.
template <int LGS>
struct Ctx {
    static constexpr int S = 1 << LGS, G = 64 >> LGS;
    static constexpr int CH = MAXCH;                                   // steps per chunk
    static constexpr int NI = 4;                                       // max (codeword, step) items per lane (CH*GW <= 256)
    int lane, g, s, GW;
    bool sr4;                                                          // 4-state shift-register trellis: static DPP exchange
    int sb[2];                                                         // sr4: MSB of next_state[s][i] (which successor input i leads to)
    bool active;                                                       // lane belongs to one of the GW decoded slots
    int nxt[2], code[2];              // outgoing branches of state s: next-state lane, 2-bit code (sys, parity)
    int plane[2], pin[2], pcode[2];   // incoming branches in np.where order: predecessor lane, input, code
    // LDS carve-up
    double *gam;    // [CH][GW][4]
    double *pr0;    // [CH][GW]
    double *lin;    // [CH][GW]   L_int of the chunk
    double *xs;     // [CH][GW*S][2] per-lane branch products a*gamma*beta of the chunk (forward pass)
    double *bt;     // [CH][GW*S]  beta rows of the chunk (forward pass)
    double *xch;    // [64]       exchange buffer
};

template <int LGS>
__device__ __forceinline__ void init_ctx(Ctx<LGS> &c, const MapTables &tb, unsigned char *smem, int GW) {
    constexpr int S = Ctx<LGS>::S, CH = Ctx<LGS>::CH;
    c.lane = threadIdx.x;
    c.GW = GW;
    c.active = (c.lane >> LGS) < GW;
    c.g = c.active ? (c.lane >> LGS) : 0;                         // idle lanes shadow slot 0 (reads only)
    c.s = c.lane & (S - 1);
    c.sr4 = (LGS == 2) && tb.sr4;
    const int gbase = (c.lane >> LGS) << LGS, sh = tb.n - 2;      // exchanges stay inside the lane's own group
    for (int i = 0; i < 2; i++) {
        c.nxt[i] = gbase + tb.next_state[c.s * 2 + i];
        c.sb[i] = (tb.next_state[c.s * 2 + i] >> (LGS - 1)) & 1;
        c.code[i] = (tb.output[c.s * 2 + i] >> sh) & 3;       // [msg_bit, parity_bit] = codeword_array[0:2] (:96-98)
        c.plane[i] = gbase + tb.pred_state[c.s * 2 + i];
        c.pin[i] = tb.pred_input[c.s * 2 + i];
        c.pcode[i] = (tb.pred_code[c.s * 2 + i] >> sh) & 3;
    }
    double *p = reinterpret_cast<double *>(smem);
    c.gam = p; p += CH * GW * 4;
    c.pr0 = p; p += CH * GW;
    c.lin = p; p += CH * GW;
    c.xs = p;  p += CH * GW * S * 2;
    c.bt = p;  p += CH * GW * S;
    c.xch = p;
}

template <int LGS>
size_t lds_bytes(int GW) {
    return sizeof(double) * (size_t)(Ctx<LGS>::CH * GW * 6 + Ctx<LGS>::CH * GW * Ctx<LGS>::S * 3 + 64);
}

constexpr int KNORM = 4;       // renormalise alpha / beta every KNORM steps

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
// the successors of state s are (s>>1) and 2|(s>>1) -- two constant quad permutes + a per-lane select instead of
// four quad broadcasts and a select tree.
template <int LGS>
__device__ __forceinline__ void exchange_succ(const Ctx<LGS> &c, double v, double &v0, double &v1) {
    if (LGS == 2 && c.sr4) {
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
template <int LGS>
__device__ __forceinline__ void exchange_pred(const Ctx<LGS> &c, double v, double &v0, double &v1) {
    if (LGS == 2 && c.sr4) {
        v0 = dppd<0x88>(v);                                       // quad_perm [0,2,0,2]
        v1 = dppd<0xDD>(v);                                       // quad_perm [1,3,1,3]
    } else {
        exchange2<LGS>(c, v, c.plane[0], c.plane[1], v0, v1);
    }
}

// ---- time-parallel stage of a chunk --------------------------------------------------------------------------
// A chunk has CH*GW (codeword, step) items, at most 4 per lane: item p = lane + 64*q -> codeword slot p / CH,
// step p % CH (consecutive lanes -> consecutive steps of one codeword: coalesced 128-byte segments).
struct RawChunk {
    double r0[4], r1[4], li[4];
};

template <int LGS>
__device__ __forceinline__ void load_raw(const Ctx<LGS> &c, RawChunk &rc, int64_t cw0, int64_t B, int64_t N, int64_t t0,
                                         int len, const double *sys, const int32_t *sys_perm, const double *par,
                                         const double *Lin, int64_t lstride) {
    constexpr int CH = Ctx<LGS>::CH;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int p = c.lane + 64 * q, gg = p / CH, tl = p % CH;
        const int64_t cw = cw0 + gg, t = t0 + tl;                 // 0-based step index
        rc.r0[q] = 0.0; rc.r1[q] = 0.0; rc.li[q] = 0.0;
        if (gg < c.GW && cw < B && tl < len) {
            rc.r0[q] = sys[cw * N + (sys_perm ? sys_perm[t] : t)];   // sys_symbols_i = interlv(sys) (turbo.py:310)
            rc.r1[q] = par[cw * N + t];
            rc.li[q] = Lin[cw * lstride + t];
        }
    }
}

// gam[tl][g][code] (_compute_branch_prob :62-76), pr0[tl][g] (priors[0] :239), lin[tl][g] into LDS
template <int LGS>
__device__ __forceinline__ void stage_chunk(const Ctx<LGS> &c, const RawChunk &rc, double nv2) {
    constexpr int CH = Ctx<LGS>::CH;
    const int GW = c.GW;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int p = c.lane + 64 * q, gg = p / CH, tl = p % CH;
        if (gg < GW) {
            double *gm = c.gam + (tl * GW + gg) * 4;
#pragma unroll
            for (int code = 0; code < 4; code++) {
                const double x = rc.r0[q] - (double)(2 * ((code >> 1) & 1) - 1);
                const double y = rc.r1[q] - (double)(2 * (code & 1) - 1);
                gm[code] = exp(-(x * x + y * y) / nv2);
            }
            c.pr0[tl * GW + gg] = 1.0 / (1.0 + exp(rc.li[q]));
            c.lin[tl * GW + gg] = rc.li[q];
        }
    }
}

// Backward recursion over one staged chunk (:78-111): beta of reference time t_lo+tl for tl = len-1 .. 0, starting
// from `b` = beta at the chunk's upper boundary.  Every row is streamed to `rows` ([CH][64] float64 in HBM, one
// coalesced 512-byte store per step, off the dependent chain): rows[tl] = beta of time t_lo + tl + 1, the row the
// forward pass needs at step tl.
template <int LGS>
__device__ __forceinline__ double beta_chunk(const Ctx<LGS> &c, double b, int len, double *__restrict__ rows) {
    const int GW = c.GW;
    if (len <= 0) return b;
    rows[(len - 1) * 64 + c.lane] = b;                            // beta[t_lo + len]: row used by the last step of the chunk
    // one step; a full chunk is unrolled (compile-time LDS offsets, operand loads hoisted by the scheduler, no
    // pointer bumps or register rotation on the dependent chain), a partial last chunk runs the same body rolled
    auto step = [&](int tl) {
        const double *gm = c.gam + (tl * GW + c.g) * 4;
        const double g0 = gm[c.code[0]], g1 = gm[c.code[1]], p0 = c.pr0[tl * GW + c.g];
        const double p1 = 1.0 - p0;                               // priors[1] = 1 - priors[0] (:240)
        double bn0, bn1;
        exchange_succ<LGS>(c, b, bn0, bn1);
        double nb = 0.0;
        nb += (bn0 * g0 * p0);                                    // (:106-108), input 0 then input 1
        nb += (bn1 * g1 * p1);
        b = nb;
        // (:110-111) every KNORM steps; any positive common factor is a valid normalisation, so the hardware
        // reciprocal (v_rcp_f64) is used instead of a correctly rounded division
        if ((tl & (KNORM - 1)) == 0) b = b * __builtin_amdgcn_rcp(group_sum<LGS>(b));
        if (tl > 0) rows[(tl - 1) * 64 + c.lane] = b;            // beta[t_lo + tl]: row used by step tl-1
    };
    if (len == Ctx<LGS>::CH) {
#pragma unroll
        for (int tl = Ctx<LGS>::CH - 1; tl >= 0; --tl) step(tl);
    } else {
        for (int tl = len - 1; tl >= 0; --tl) step(tl);
    }
    return b;
}

// One MAP pass over the GW codewords of this wavefront.  ckpt: [nchunks][CH][64] beta rows of this wave (HBM).
// Lout (stride lstride per codeword) receives L_int + log(app1/app0).
template <int LGS>
__device__ void map_pass(const Ctx<LGS> &c, int64_t cw0, int64_t B, int64_t N, double nv2, const double *sys,
                         const int32_t *sys_perm, const double *par, const double *Lin, int64_t lstride,
                         double *ckpt, double *Lout) {
    constexpr int S = Ctx<LGS>::S, CH = Ctx<LGS>::CH;
    const int GW = c.GW, W = GW * S, col = c.g * S + c.s;
    const int nx0 = c.nxt[0] & (W - 1), nx1 = c.nxt[1] & (W - 1);   // next-state columns inside the bt rows
    const int64_t nchunks = (N + CH - 1) / CH;
    RawChunk cur, nxt;
    // ---------------- backward pass: chunks from the end, beta rows streamed to HBM ----------------
    double b = 1.0;                                               // b_state_metrics[:, N] = 1 (:225)
    {
        const int64_t t_lo = (nchunks - 1) * CH;
        load_raw<LGS>(c, cur, cw0, B, N, t_lo, (int)(N - t_lo), sys, sys_perm, par, Lin, lstride);
    }
    for (int64_t k = nchunks - 1; k >= 0; --k) {                  // chunk k = 0-based steps [k*CH, min(N, (k+1)*CH))
        const int64_t t_lo = k * CH;
        const int len = (int)((N - t_lo < CH) ? (N - t_lo) : CH);
        __syncthreads();
        stage_chunk<LGS>(c, cur, nv2);
        if (k > 0) load_raw<LGS>(c, nxt, cw0, B, N, t_lo - CH, CH, sys, sys_perm, par, Lin, lstride);   // prefetch
        __syncthreads();
        b = beta_chunk<LGS>(c, b, len, ckpt + k * CH * 64);
        cur = nxt;
    }
    __syncthreads();
    // ---------------- forward pass: the chunk's beta rows come back from HBM (prefetched one chunk ahead),
    // then alpha + LLR (:114-158) ----------------
    double a = (c.s == 0) ? 1.0 : 0.0;                            // f_state_metrics[0][0] = 1 (:221)
    load_raw<LGS>(c, cur, cw0, B, N, 0, (int)((N < CH) ? N : CH), sys, sys_perm, par, Lin, lstride);
    double brow[CH], brown[CH];
#pragma unroll
    for (int tl = 0; tl < CH; tl++) brow[tl] = ckpt[tl * 64 + c.lane];
    for (int64_t k = 0; k < nchunks; ++k) {
        const int64_t t_lo = k * CH;
        const int len = (int)((N - t_lo < CH) ? (N - t_lo) : CH);
        __syncthreads();
        stage_chunk<LGS>(c, cur, nv2);
        if (c.active) {
#pragma unroll
            for (int tl = 0; tl < CH; tl++) c.bt[tl * W + col] = brow[tl];   // bt[tl] = beta[t_lo + tl + 1]
        }
        if (k + 1 < nchunks) {
            const int64_t t2 = t_lo + CH;
            load_raw<LGS>(c, nxt, cw0, B, N, t2, (int)((N - t2 < CH) ? (N - t2) : CH), sys, sys_perm, par, Lin, lstride);
#pragma unroll
            for (int tl = 0; tl < CH; tl++) brown[tl] = ckpt[((k + 1) * CH + tl) * 64 + c.lane];
        }
        __syncthreads();
        {
            auto step = [&](int tl) {
                const double *gm = c.gam + (tl * GW + c.g) * 4;
                const double go0 = gm[c.code[0]], go1 = gm[c.code[1]], gi0 = gm[c.pcode[0]], gi1 = gm[c.pcode[1]];
                const double p0 = c.pr0[tl * GW + c.g], p1 = 1.0 - p0;
                const double bt0 = c.bt[tl * W + c.g * S + (nx0 & (S - 1))], bt1 = c.bt[tl * W + c.g * S + (nx1 & (S - 1))];
                // app[i] += f[cs,0] * branch_prob * b[next_state, t]   (:141-143): products parked, summed in the epilogue
                double2 xv;
                xv.x = a * go0 * bt0;
                xv.y = a * go1 * bt1;
                if (c.active) *reinterpret_cast<double2 *>(c.xs + (tl * W + col) * 2) = xv;
                // f[next,1] += f[cs,0] * branch_prob * priors[input]     (:136-138), accumulation in (cs, input) order
                double ap0, ap1;
                exchange_pred<LGS>(c, a, ap0, ap1);
                double na = 0.0;
                na += (ap0 * gi0 * (c.pin[0] ? p1 : p0));
                na += (ap1 * gi1 * (c.pin[1] ? p1 : p0));
                a = na;
                if ((tl & (KNORM - 1)) == KNORM - 1) a = a * __builtin_amdgcn_rcp(group_sum<LGS>(a));   // (:155-158), every KNORM steps
            };
            if (len == CH) {
#pragma unroll
                for (int tl = 0; tl < CH; tl++) step(tl);
            } else {
                for (int tl = 0; tl < len; tl++) step(tl);
            }
        }
        __syncthreads();
        // time-parallel epilogue of the chunk: app sums in state order, L = L_int + log(app1/app0) (:145)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int p = c.lane + 64 * q, gg = p / CH, tl = p % CH;
            const int64_t cw = cw0 + gg;
            if (gg < GW && cw < B && tl < len) {
                const double *x = c.xs + (tl * W + gg * S) * 2;
                double app0 = 0.0, app1 = 0.0;
#pragma unroll
                for (int st = 0; st < S; st++) { app0 += x[2 * st]; app1 += x[2 * st + 1]; }
                Lout[cw * lstride + t_lo + tl] = c.lin[tl * GW + gg] + fast_log(app1 / app0);
            }
        }
        cur = nxt;
#pragma unroll
        for (int tl = 0; tl < CH; tl++) brow[tl] = brown[tl];
    }
    __syncthreads();
}

struct MapParams {
    MapTables tb;
    const double *sys, *par, *Lin;     // [B][N]
    double *Lout;                      // [B][N]
    uint8_t *bits;                     // [B][N]
    double *scratch;                   // per wave: beta rows [nchunks][CH][64]
    int64_t B, N;
    double nv2;
    int want_bits, GW;
};

template <int LGS>
__global__ __launch_bounds__(64) void map_decode_kernel(MapParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Ctx<LGS> c;
    init_ctx<LGS>(c, p.tb, smem, p.GW);
    const int64_t cw0 = (int64_t)blockIdx.x * p.GW;
    const int64_t wslab = ((p.N + Ctx<LGS>::CH - 1) / Ctx<LGS>::CH) * Ctx<LGS>::CH * 64;
    double *ckpt = p.scratch + (int64_t)blockIdx.x * wslab;
    map_pass<LGS>(c, cw0, p.B, p.N, p.nv2, p.sys, nullptr, p.par, p.Lin, p.N, ckpt, p.Lout);
    const int64_t cw = cw0 + c.g;
    if (c.active && cw < p.B)
        for (int64_t t = c.s; t < p.N; t += Ctx<LGS>::S)           // decoded_bits: L > 0 in 'decode' mode only (:148-152)
            p.bits[cw * p.N + t] = (uint8_t)((p.want_bits && p.Lout[cw * p.N + t] > 0) ? 1 : 0);
}

struct TurboParams {
    MapTables tb;
    const double *sys, *p1, *p2, *Lint;   // [B][N], Lint may be null
    const int32_t *perm;                  // [N]
    uint8_t *bits;                        // [B][N]
    double *beta;                         // per wave: beta rows [nchunks][CH][64]
    double *larr;                         // per codeword: A[N] B[N] C[N]
    int64_t B, N;
    double nv2;
    int n_iter, GW;
};

template <int LGS>
__global__ __launch_bounds__(64) void turbo_decode_kernel(TurboParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Ctx<LGS> c;
    init_ctx<LGS>(c, p.tb, smem, p.GW);
    constexpr int S = Ctx<LGS>::S;
    const int64_t cw0 = (int64_t)blockIdx.x * p.GW, N = p.N;
    const int64_t cw = cw0 + c.g;
    const bool valid = c.active && cw < p.B;
    const int64_t wslab = ((N + Ctx<LGS>::CH - 1) / Ctx<LGS>::CH) * Ctx<LGS>::CH * 64;
    double *beta = p.beta + (int64_t)blockIdx.x * wslab;
    // L arrays of all codewords: [B][3][N] -> per-codeword stride 3N; A at +0, B at +N, C at +2N
    double *A0 = p.larr, *B0 = p.larr + N, *C0 = p.larr + 2 * N;
    const int64_t ls = 3 * N;
    // The element-wise stages between the MAP passes are done by the whole wave, one codeword after the other, 64
    // consecutive elements per instruction and four independent rounds in flight.  (The first version gave every
    // codeword to its own S lanes: N/S dependent load -> store rounds per stage, 3.9 of the 14.0 ms of config 3.)
    const int lane = c.lane, GW = p.GW;
    (void)cw; (void)valid; (void)S;
    for (int g = 0; g < GW; g++) {
        const int64_t cwg = cw0 + g;
        if (cwg >= p.B) break;
        double *A = A0 + cwg * ls;
#pragma unroll 4
        for (int64_t t = lane; t < N; t += 64) A[t] = p.Lint ? p.Lint[cwg * N + t] : 0.0;      // L_int_1 (:305-308)
    }
    __syncthreads();
    for (int it = 0; it < p.n_iter; it++) {
        // [L_ext_1, _] = map_decode(sys, non_sys_1, trellis, nv, L_int_1, 'compute')   (:315)
        map_pass<LGS>(c, cw0, p.B, N, p.nv2, p.sys, nullptr, p.p1, A0, ls, beta, B0);
        __syncthreads();
        // L_ext_1 -= L_int_1 ; L_int_2 = interlv(L_ext_1)                               (:318-319), one pass: same subtraction
        for (int g = 0; g < GW; g++) {
            const int64_t cwg = cw0 + g;
            if (cwg >= p.B) break;
            const double *A = A0 + cwg * ls, *Bb = B0 + cwg * ls;
            double *C = C0 + cwg * ls;
#pragma unroll 4
            for (int64_t t = lane; t < N; t += 64) {
                const int32_t q = p.perm[t];
                C[t] = Bb[q] - A[q];
            }
        }
        __syncthreads();
        // [L_2, bits] = map_decode(sys_i, non_sys_2, trellis, nv, L_int_2, mode)          (:326)
        map_pass<LGS>(c, cw0, p.B, N, p.nv2, p.sys, p.perm, p.p2, C0, ls, beta, B0);
        __syncthreads();
        // L_ext_2 = L_2 - L_int_2 ; L_int_1 = deinterlv(L_ext_2)                          (:328-329)
        for (int g = 0; g < GW; g++) {
            const int64_t cwg = cw0 + g;
            if (cwg >= p.B) break;
            const double *Bb = B0 + cwg * ls, *C = C0 + cwg * ls;
            double *A = A0 + cwg * ls;
#pragma unroll 4
            for (int64_t t = lane; t < N; t += 64) A[p.perm[t]] = Bb[t] - C[t];
        }
        __syncthreads();
    }
    // decoded_bits = deinterlv(decoded_bits of the last MAP2)                              (:331)
    for (int g = 0; g < GW; g++) {
        const int64_t cwg = cw0 + g;
        if (cwg >= p.B) break;
        const double *Bb = B0 + cwg * ls;
#pragma unroll 4
        for (int64_t t = lane; t < N; t += 64)
            p.bits[cwg * N + p.perm[t]] = (uint8_t)((p.n_iter > 0 && Bb[t] > 0) ? 1 : 0);
    }
}

// codewords per wavefront: full wavefronts as soon as the batch gives every SIMD of the chip one of them; for
// smaller batches fewer codewords per wavefront (idle lanes) spread the work over more SIMDs.  (Measured on
// MI355X, B = 16384 x N = 1024: GW = 16 -> 1.07 ms per MAP pass, GW = 4 -> 1.78 ms: the pass is bound by
// instruction issue, not by latency, so idle lanes do not pay once the chip is full.)
int pick_gw(int S, int64_t B) {
    int G = 64 / S;
    if (G > 16) G = 16;                                            // CH*GW <= 256 items per chunk
    int gw = G;
    while (gw > 1 && (B + gw - 1) / gw < 1024) gw >>= 1;
    return gw;
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
    const int64_t nblocks = (B + GW - 1) / GW;
    p.GW = GW;
    p.sys = d_sys; p.par = d_par; p.Lin = d_L_int; p.Lout = d_L_ext; p.bits = d_bits;
    p.B = B; p.N = N; p.nv2 = 2 * noise_variance; p.want_bits = want_bits;
    if ((rc = workspace(st, 0, sizeof(double) * (size_t)(nblocks * ((N + MAXCH - 1) / MAXCH) * MAXCH * 64), (void **)&p.scratch))) return rc;
    dim3 grid((unsigned)nblocks), block(64);
    switch (p.tb.lgS) {
#define CASE(LG) case LG: hipLaunchKernelGGL(map_decode_kernel<LG>, grid, block, lds_bytes<LG>(GW), st, p); break;
        CASE(1) CASE(2) CASE(3) CASE(4)
#undef CASE
        default: set_error("map_decode: unsupported state count"); return CPX_ELIMIT;
    }
    CPX_HIP(hipGetLastError());
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
    const int64_t nblocks = (B + GW - 1) / GW;
    p.GW = GW;
    p.sys = d_sys; p.p1 = d_p1; p.p2 = d_p2; p.Lint = d_L_int_or_null; p.perm = d_perm; p.bits = d_bits;
    p.B = B; p.N = N; p.nv2 = 2 * noise_variance; p.n_iter = n_iter;
    if ((rc = workspace(st, 0, sizeof(double) * (size_t)(nblocks * ((N + MAXCH - 1) / MAXCH) * MAXCH * 64), (void **)&p.beta))) return rc;
    if ((rc = workspace(st, 1, sizeof(double) * (size_t)(B * 3 * N), (void **)&p.larr))) return rc;
    dim3 grid((unsigned)nblocks), block(64);
    switch (p.tb.lgS) {
#define CASE(LG) case LG: hipLaunchKernelGGL(turbo_decode_kernel<LG>, grid, block, lds_bytes<LG>(GW), st, p); break;
        CASE(1) CASE(2) CASE(3) CASE(4)
#undef CASE
        default: set_error("turbo_decode: unsupported state count"); return CPX_ELIMIT;
    }
    CPX_HIP(hipGetLastError());
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
