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
//   phase 2   F continues alpha over the second half and R continues beta over the first half.  Per chunk a wave runs
//             alpha over the chunk (values kept in registers), then beta over it together with the a-posteriori
//             products; one of the two is the wave's own chain, the other is RECOMPUTED from the partner's checkpoint
//             (same operations) and dropped.
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
// The sum-normalisation of the reference (turbo.py:110-111, :155-158) is applied every KNORM = 8 steps with the hardware
// reciprocal: a common positive factor per time step cancels in app1/app0 and in every later normalisation, so LLRs
// differ from the reference's only by rounding (measured <= 1e-13; tolerance 1e-5).
// Waves of a workgroup only meet at the phase boundary and between the stages of turbo_decode (__syncthreads +
// workgroup-scope fences: the partner's checkpoints are read through the CU's L1 / L2).  turbo_decode is a SEQUENCE of launches
// (since round 3): turbo_init_kernel (slab), one turbo_pass_kernel per MAP pass -- since round 6 the slab is time-major and the
// interleaver / deinterleaver are the ROW INDEX a pass reads its prior through -- and turbo_final_kernel (decisions); see TurboParams.
#include "cpx_internal.h"
#include "cpx_math.h"

#include <cstdlib>
#include <type_traits>

using namespace cpx;

namespace {

constexpr int CH = 8;          // steps per chunk
#ifndef CPX_KNORM
#define CPX_KNORM 8
#endif
constexpr int KNORM = CPX_KNORM;   // renormalise alpha / beta every KNORM steps (a power of two <= CH).  Round 4: 8 = once per chunk (4: +1 % on a
                                   // turbo decode, +2.4 % on a MAP pass; 2: +3 % / +8 %).  Every step multiplies by gamma' p <= 1, so the sums only
                                   // shrink between two normalisations and flag (C) -- a sum below 1e-150 -- bounds all of them whatever the interval.
// ---- LDS layout (round 4: re-laid for zero bank conflicts; scripts/micro/lds_bank_sim.py is the model it was chosen with) ----
// rocprofv3 had SQ_LDS_BANK_CONFLICT at 2.0 cycles per LDS instruction in rounds 2 and 3.  The model, with the lane groups and
// bank functions of the CDNA4 LDS (ds_read_b64: 2 x 32 lanes over 64 dword banks, ds_read_b128: 4 x 16 interleaved lanes,
// ds_write_b64: 4 x 16 lanes over 32 banks), found three sources in the round-3 layout ([step][codeword][gamma x 4, prior x 2],
// products [step][lane][2], rows of 132 doubles): the 6-double item stride puts the gamma of codewords g and g + 5 on one bank
// (2 extra cycles per gamma read), a lane's two products are adjacent so lanes l and l + 8 of a 16-lane store group collide
// (4 per store), and the epilogue's 16-lane b128 groups mix steps 0-3 of two codewords with steps 4-7 of two others (12 per read).
// Now: gamma [step][codeword][4] (row = 4 GW + 2 doubles) and priors [step][codeword][2] (row = 2 GW + 2) are separate tables --
// 32 consecutive lanes read 32 different 8-byte slots --, the products are slot-major [step][input][lane] (row = 130), and the
// (codeword, step) item of a lane is permuted so that every b128 lane group holds ONE half (steps 0-3 or 4-7) of four codewords
// of equal parity (item_of).  Model: 0 conflict cycles on all 94 LDS instructions of a phase-2 chunk (round 3: 224).
constexpr int XS_ROW = 2 * 64 + 2;   // doubles per step row of the parked branch products: [input 0 | input 1][64 lanes] + 2 pad
constexpr int NPAIR = 4;       // (forward, reverse) wave pairs per full-size workgroup: 8 waves = 2 per SIMD of a CU
constexpr unsigned TM_RW = 16; // codeword slots per row of turbo_decode's time-major slab (TurboParams): one 128-byte line of float64

// ---- outside the reference's representable range: detect and redo --------------------------------------------------------
// The recursions below are SCALE-FREE in the branch probabilities (a factor common to a trellis step is dropped) and
// normalise every KNORM steps; the reference carries the absolute gamma = exp(-d^2 / 2 sigma^2) and normalises every step
// (turbo.py:62-76, :110-111, :155-158).  Wherever every term of the reference stays a normal float64 the two agree to
// rounding; where its terms underflow (symbol amplitudes of 5 - 20 at sigma^2 <= 0.1, priors of e^-200 meeting a contradicting
// channel) the reference returns NaN / +-inf LLRs and these kernels would not.  So the fast kernels DETECT, conservatively,
// every codeword for which that can happen and a literal absolute-scale kernel (bcjr_exact.hip) decodes those again:
//   (A) a received pair whose worst branch probability is below e^-345:  (|r0| + 1)^2 + (|r1| + 1)^2 > 345 * 2 sigma^2
//       (also true for NaN / inf);
//   (C) a normalisation sum of alpha or beta below 1e-150 (or NaN): up to KNORM steps lost 150 decades;
//   (D) a non-finite LLR;  (E) both a-posteriori sums of a step below 1e-150.
// (A) bounds what the dropped common factor can be (>= 1e-150 per step), (C) / (E) what the scale-free quantities are, so a
// codeword that raises none of them has every term of the reference's own recursion above 1e-300.  The priors are NOT
// scale-free: p0 = 1 / (1 + e^L), p1 = 1 - p0 as the reference computes them (:239-240), exact zeros and cancellation included.
constexpr double T_A = 345.0, T_SMALL = 1e-150;

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

// NumPy's float64 add.reduce over the S state values of a codeword (turbo.py:110-111, :155-156), evaluated in EVERY lane of the
// group -- the order matters to the literal kernel below: fewer than 8 values are added sequentially from index 0, eight or more
// through eight accumulators r[j] = a[j] (+ a[j + 8]) combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) (bcjr_exact.hip np_sum_leaf).
// IEEE addition is commutative, so a butterfly whose partners are the tree's siblings gives every lane that very value.
template <int LGS>
__device__ __forceinline__ double np_group_sum(double v) {
    if (LGS == 1) return v + dppd<0xB1>(v);                       // a0 + a1
    if (LGS == 2) {                                               // ((a0 + a1) + a2) + a3: quad broadcasts
        const double q0 = dppd<0x00>(v), q1 = dppd<0x55>(v), q2 = dppd<0xAA>(v), q3 = dppd<0xFF>(v);
        return ((q0 + q1) + q2) + q3;
    }
    if (LGS == 4) v += dppd<0x128>(v);                            // row_ror:8 -- r[j] = a[j] + a[j + 8]
    v += dppd<0xB1>(v);                                           // r0 + r1, r2 + r3, ...
    v += dppd<0x4E>(v);                                           // (r0 + r1) + (r2 + r3), ...
    v += dppd<0x141>(v);                                          // row_half_mirror: the two quads of eight lanes
    return v;
}

// One wavefront of a pair.  GW = codewords decoded by the pair (power of two, <= 64/S and <= 16): when the batch
// cannot fill the chip the host picks GW < 64/S (idle lanes, more pairs).
template <int LGS>
struct Ctx {
    static constexpr int S = 1 << LGS;
    static constexpr int NI = 2;                                       // (codeword, step) items per lane: CH*GW <= 128
    int lane, g, s, GW, GS, PS;                                        // GS / PS: doubles per step row of `gam` / `pri`
    bool fwd;                                                          // forward (alpha) wave of the pair
    int sb[2];                                                         // sr4: MSB of next_state[s][i] (which successor input i leads to)
    bool active;                                                       // lane belongs to one of the GW decoded slots
    int nxt[2], code[2];              // outgoing branches of state s: next-state lane, 2-bit code (sys, parity)
    int plane[2], pin[2], pcode[2];   // incoming branches in np.where order: predecessor lane, input, code
    // 4-state shift-register trellises: the successors of s are lo = s>>1 and hi = 2|(s>>1) whatever the input; ilo is
    // the input that leads to lo.  Everything a step needs is then a STATIC per-lane offset -- no selects on the chain.
    int ilo;
    int o_glo, o_ghi;                 // offsets of gamma(code of the branch to lo / hi) inside an item of `tab`
    // LDS of this wave
    double *gam;    // [CH][GW][4] (+2 pad per step): gamma[4] of every (step, codeword) item
    double *pri;    // [CH][GW][2] (+2 pad per step): priors p0, p1
    double *xs;     // [CH][2][64] (+2 pad per step): per-lane branch products alpha*gamma*beta of the chunk, by input (phase 2)
    double *rw;     // [CH][64]     alpha rows of a partial (last) chunk: the rolled code path keeps them here
    double *xch;    // [64]         exchange buffer (trellises without the DPP fast path)
    // "detect and redo": lanes that raised a flag, one mask per lane -> codeword mapping so that a flag sends ITS codeword to the
    // redo path and not all GW of the pair (round 4): bad_s -- flags of the recursions, lane = (codeword slot, state);
    // bad_i[q] -- flags of the time-parallel stage / epilogue, lane = item q of item_of.  mutable: the helpers take the context by
    // const reference
    mutable unsigned long long bad_s, bad_i0, bad_i1;
};

// v is wave-level: OR the lanes for which `cond` holds into a mask on the scalar unit
__device__ __forceinline__ void flag_or(unsigned long long &mask, bool cond) { mask |= __ballot(cond); }

template <int LGS>
__host__ __device__ constexpr size_t wave_lds_doubles(int GW) { return (size_t)CH * (GW * 6 + 4) + (size_t)CH * (XS_ROW + 64) + 64; }

// The (codeword slot, step) item a lane stages, loads and finishes (two per lane, q = 0 / 1: CH * GW <= 128 items per chunk).
// Eight consecutive lanes hold the eight steps of ONE codeword (64-byte segments of the pass's arrays); which codeword and in
// which order is chosen for the epilogue's ds_read_b128 lane groups {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} (+32): blocks
// 1, 2, 5, 6 hold their steps with the halves swapped, blocks 0-3 serve the even codewords and blocks 4-7 the odd ones.
__device__ __forceinline__ void item_of(int lane, int q, int &gg, int &tl) {
    const int bk = lane >> 3;
    gg = (((bk & 3) << 1) | (bk >> 2)) + 8 * q;
    tl = (lane & 7) ^ ((((bk + 1) >> 1) & 1) << 2);
}
// TM (turbo_decode, round 6): the slab between the passes is TIME-MAJOR -- per array N rows of 16 codeword slots (TurboParams) -- so
// eight consecutive lanes hold eight consecutive slots of ONE step (a 64-byte half row), a lane's two items are slots g and g + 8 of
// that step and a chunk of an array is eight whole rows: the rows of a step can then sit ANYWHERE in the array, which is what lets
// the interleaver disappear into the row index a pass reads through (turbo_pass_kernel).  One item per lane for GW <= 8, like
// item_of.  (Adjacent slots 2 g, 2 g + 1 per lane and 16-byte accesses were built first: same time for 4 states -- 3.27 against
// 3.28 ms per config-3 decode --, half the lanes idle in the stage / epilogue for 8 states: 7.39 against 6.90 ms; and the compiler
// scheduled the VALU that overwrites its data straight behind `buffer_store_dwordx4 ..., s47 offen`, a hazard it only guards for a
// constant scalar offset: lanes 4-7 of every eight stored the wrong first slot.  experiments/README.md.)
template <bool TM>
__device__ __forceinline__ void item_map(int lane, int q, int &gg, int &tl) {
    if (TM) {
        gg = (lane & 7) + 8 * q;
        tl = lane >> 3;
    } else {
        item_of(lane, q, gg, tl);
    }
}

template <int LGS>
__device__ __forceinline__ void init_ctx(Ctx<LGS> &c, const MapTables &tb, unsigned char *smem, int GW) {
    constexpr int S = Ctx<LGS>::S;
    const int wave = threadIdx.x >> 6;
    c.lane = threadIdx.x & 63;
    c.fwd = (wave & 1) == 0;
    c.GW = GW;
    c.GS = GW * 4 + 2;
    c.PS = GW * 2 + 2;
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
    c.ilo = c.sb[0] ? 1 : 0;
    c.o_glo = c.sb[0] ? c.code[1] : c.code[0];
    c.o_ghi = c.sb[0] ? c.code[0] : c.code[1];
    double *p = reinterpret_cast<double *>(smem) + (size_t)wave * wave_lds_doubles<LGS>(GW);
    c.gam = p; p += CH * c.GS;
    c.pri = p; p += CH * c.PS;
    c.xs = p;  p += CH * XS_ROW;
    c.rw = p;  p += CH * 64;
    c.xch = p;
    c.bad_s = 0; c.bad_i0 = 0; c.bad_i1 = 0;
}

// flags raised during the pass -> flag bytes of the codewords they belong to (both waves of a pair may store the same 1): lane cw
// looks up the lanes of its own codeword slot in the three masks
template <int LGS, bool TM = false>
__device__ __forceinline__ void publish_flags(const Ctx<LGS> &c, uint8_t *flags, int ncw) {
    constexpr int S = Ctx<LGS>::S;
    if (flags && (c.bad_s | c.bad_i0 | c.bad_i1) != 0 && c.lane < ncw) {
        const int cw = c.lane;
        const unsigned long long st = (c.bad_s >> (cw * S)) & ((S == 64) ? ~0ull : ((1ull << S) - 1ull));
        unsigned long long it;
        if (TM) {                                                     // item_map<true>: lanes 8 tl + (cw & 7), q = cw >> 3
            it = (((cw >> 3) ? c.bad_i1 : c.bad_i0) >> (cw & 7)) & 0x0101010101010101ull;
        } else {
            const int gg = cw & 7, bk = ((gg & 1) << 2) | (gg >> 1);  // item_of: block bk of eight lanes serves codeword slot gg (+ 8 q)
            it = (((cw >> 3) ? c.bad_i1 : c.bad_i0) >> (8 * bk)) & 0xffull;
        }
        if ((st | it) != 0) flags[cw] = 1;
    }
    c.bad_s = 0; c.bad_i0 = 0; c.bad_i1 = 0;
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
// A chunk has CH*GW (codeword, step) items, at most 2 per lane: item_of(lane, q) -> (codeword slot, step); eight consecutive
// lanes hold the eight steps of one codeword: 64-byte segments of the pass's arrays.
struct RawChunk {
    double r0[2], r1[2], li[2];
    unsigned ridx;                // TM: the row (of the prior's array, and of the systematic array if it is gathered) of this lane's
                                  // step in the chunk that will be loaded into this set NEXT (requested two chunks ahead of its use)
};

// The arrays of a pass, seen from the pair of wavefronts that runs it: every pointer is wave-uniform and already points at
// the pair's first codeword; a lane adds a 32-bit BYTE offset (codeword-in-pair * stride + step).  With 64-bit per-lane
// addresses the pass kept ten address pairs in VGPRs and the (round-2, single-launch) turbo kernel spilled them: every reload is a scratch load
// that retires through the same in-order counter as the prefetches, i.e. it waits for the HBM round trip issued just before
// it (the "exposed memory time" of the round-2 ablations).
struct PassIO {
    // The four arrays of a pass as RAW BUFFERS that start at the pair's first codeword and are 2 GiB long; a lane's byte offset
    // is (codeword-in-pair * stride + step) * 8 < 2^31 (checked on the host), and a lane that has nothing to load or store uses
    // offset OOB = 2^31: the hardware returns 0 for that load and drops that store.  No branch around a memory operation
    // is left in the pass, and that is the point: with `if (in range) load` the compiler cannot know how many operations are in
    // flight behind the checkpoint load that starts a recursion, so it waited for ALL of them (s_waitcnt vmcnt(0)) -- i.e. for
    // the HBM round trip of the prefetch issued just before, once per chunk.  With unconditional operations it counts.
    __amdgpu_buffer_rsrc_t rsys, rpar, rlin, rout;   // systematic, parity, L_int, output (L_int + log(app1/app0), or with `ext` the log alone)
    __amdgpu_buffer_rsrc_t rbits;                    // map_decode: hard decisions [.][N] (L > 0 in 'decode' mode); zero-length if unwanted
    unsigned osys, opar, olin, oout;                 // scalar byte offsets added to the lane offsets (turbo: the array inside the slab)
    int want_bits;
    int sstride, pstride;         // per-codeword strides (elements)
    bool ext;                     // turbo: write L - L_int, the quantity the next half-iteration interleaves (:318, :328)
    bool pout;                    // turbo, all passes but the last two: write prior0(L - L_int) = app0 / (app0 + app1) instead (see epilogue)
    int lstride, ncw, N;          // ncw: codewords of the batch this pair really has (<= GW, may be <= 0)
    double nv2;
    double *ckpt;                 // this pair's checkpoint rows [nchunks + 1][64]
    // TM (turbo_pass_kernel): rows of TM_RW slots, ALL arrays through rsys (one descriptor: a second and a third one pushed the kernel
    // over its scalar registers -- 36 spilled, 72 v_readlane / v_writelane in the loops, the 8-state pass 476 -> 567 us); idx = the row
    // index table of this pass (interleaver, its inverse or the identity: the prior's row of step t is idx[t], N entries), sysg = all
    // ones when the systematic factors are gathered through it too (MAP 2) else 0; llr_in: the prior array holds LLRs, not P(bit = 0)
    unsigned sysg;
    unsigned odec;                // scalar byte offset of this pair's decision bytes
    unsigned st_out, st_dec;      // all ones / 0: this pass stores its output array / its decisions (the last MAP 2 stores only those)
    const int32_t *idx;
    bool llr_in;
    uint8_t *flags;               // "detect and redo": one byte per codeword of the pair, set to 1 (never cleared here); may be null
    bool abort_ok;                // map_decode (round 5): its redo launch decodes a WHOLE pair again as soon as one codeword of it is flagged,
                                  // so a wave that has raised a flag may stop working (its pair's outputs are garbage until then)
};
constexpr unsigned OOB = 0x80000000u;
typedef unsigned bcjr_v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pass_buffer(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}
// all ones if this lane's item (codeword slot gg, step tl of the chunk) exists, else 0 -- plain integer arithmetic: from
// `cond ? offset : OOB` the compiler built a branch with the loads inside, which is the uncounted wait again
__device__ __forceinline__ unsigned item_mask(int gg, int ncw, int tl, int len) {
    return (unsigned)(((gg - ncw) & (tl - len)) >> 31);
}
__device__ __forceinline__ unsigned item_off(unsigned m, unsigned byte_off) { return (byte_off & m) | (OOB & ~m); }
__device__ __forceinline__ double buf_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_st(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(bcjr_v2u, v), r, voff, soff, 0);
}

// ---- "fp32-fast" turbo decoding (cpx_set_precision; NOT the parity mode; round 5) ---------------------------------------------------
// The arithmetic of a pass stays float64; what changes is the SLAB between the passes: float32 arrays, so that a pass reads 12 instead
// of 24 bytes per step and wave and the interleave stages move half the bytes.  A prior cannot be stored as p0 in float32 (1 - p0
// would lose everything below 6e-8: every |L| > 16.6 would become a hard decision), so the slab carries the SMALLER of (p0, p1) with
// the sign bit saying which: q = +p1 when p1 <= 1/2, q = -p0 otherwise; the other one is 1 - |q|.  Channel factors and the LLRs of the
// last two passes are plain float32.  Contract: tests/test_fp32_fast_gpu.py (bit error rate and fraction of differing bits against
// the float64 decode), DESIGN 4.2.
template <bool S32>
__device__ __forceinline__ double slab_ld(__amdgpu_buffer_rsrc_t r, unsigned m, unsigned elem, unsigned soff) {
    if (S32) return (double)__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, item_off(m, elem * 4u), soff, 0));
    return buf_ld(r, item_off(m, elem * 8u), soff);
}
template <bool S32>
__device__ __forceinline__ void slab_st(__amdgpu_buffer_rsrc_t r, unsigned m, unsigned elem, unsigned soff, double v) {
    if (S32) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)v), r, item_off(m, elem * 4u), soff, 0);
    else buf_st(r, item_off(m, elem * 8u), soff, v);
}
// float32 slab: the prior of odds r = P1 / P0 (p0 = 1 / (1 + r); the smaller of the two, signed)
__device__ __forceinline__ double enc_prior_odds(double r) {
    const double p0 = 1.0 / (1.0 + r);
    return r <= 1.0 ? r * p0 : -p0;
}
__device__ __forceinline__ void dec_prior(double q, double &p0, double &p1) {
    const double v = fabs(q), w = 1.0 - v;
    const bool neg = __double2hiint(q) < 0;                       // the sign BIT: -0.0 means p0 = 0
    p0 = neg ? v : w;
    p1 = neg ? w : v;
}

__device__ __forceinline__ double ld_off(const double *base, unsigned elem) {
    return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + (elem * 8u));
}
__device__ __forceinline__ void st_off(double *base, unsigned elem, double v) {
    *reinterpret_cast<double *>(reinterpret_cast<char *>(base) + (elem * 8u)) = v;
}

// TM: requests the chunk [t0, t0 + len) into `rc` using the row index rc.ridx holds for it, then requests the row index of the chunk
// [tn, ...) this set will be loaded with next time -- two loads ahead, so neither wait is ever on the critical path
template <int LGS, bool S32 = false, bool TM = false>
__device__ __forceinline__ void load_raw(const Ctx<LGS> &c, const PassIO &io, RawChunk &rc, int t0, int len, int tn = 0) {
    if (TM) {
        const int tl = c.lane >> 3, g0 = c.lane & 7;
        const unsigned t = (unsigned)(t0 + tl);
        const unsigned row = rc.ridx;
        const unsigned srow = (row & io.sysg) | (t & ~io.sysg);
#pragma unroll
        for (int q = 0; q < 2; q++) {                                 // slots g0 and g0 + 8: the two 64-byte halves of the rows
            const int gg = g0 + 8 * q;
            const unsigned m = item_mask(gg, c.GW, tl, len);
            rc.r0[q] = slab_ld<S32>(io.rsys, m, srow * TM_RW + gg, io.osys);
            rc.r1[q] = slab_ld<S32>(io.rpar, m, t * TM_RW + gg, io.opar);
            rc.li[q] = slab_ld<S32>(io.rlin, m, row * TM_RW + gg, io.olin);
        }
        const int tq = tn + tl;
        rc.ridx = (unsigned)io.idx[tq < io.N ? tq : io.N - 1];        // (past the end of a partial last chunk: any valid row)
        return;
    }
#pragma unroll
    for (int q = 0; q < 2; q++) {
        int gg, tl;
        item_of(c.lane, q, gg, tl);
        const unsigned t = (unsigned)(t0 + tl);                   // 0-based step index
        const unsigned m = item_mask(gg, io.ncw, tl, len);        // no such item: zeros
        rc.r0[q] = slab_ld<S32>(io.rsys, m, (unsigned)(gg * io.sstride) + t, io.osys);
        rc.r1[q] = slab_ld<S32>(io.rpar, m, (unsigned)(gg * io.pstride) + t, io.opar);
        rc.li[q] = slab_ld<S32>(io.rlin, m, (unsigned)(gg * io.lstride) + t, io.olin);
    }
}

// gam[tl][g] = gamma'[0..3], pri[tl][g] = q0, q1 into LDS -- the branch probabilities (_compute_branch_prob :62-76) and the priors
// (:239-240) of every item, each up to a factor that is COMMON to the step and therefore cancels in the normalised
// recursions and in app1/app0:
//   gamma[c] = exp(-((r0-c0)^2 + (r1-c1)^2)/nv2) = E * (c0 matches sign(r0) ? 1 : Qa) * (c1 matches sign(r1) ? 1 : Qb),
//     E = exp(-((|r0|-1)^2 + (|r1|-1)^2)/nv2) (dropped), Qa = exp(-4|r0|/nv2), Qb = exp(-4|r1|/nv2): two exps instead
//     of four, every factor <= 1;
//   (p0, p1) = (1 / (1 + e^L), 1 - p0): the reference's own two operations (round 2 used the scale-free weights (1, e^L) /
//     (e^-L, 1); they are more accurate than the reference where 1 - p0 cancels -- L < -30 -- and non-zero where its p1 is
//     exactly 0 -- L < -36.7 --, which is a difference as soon as the channel contradicts such a prior).
// Row strides of 4 GW + 2 / 2 GW + 2 doubles: the eight lanes that hold the steps of a codeword store to eight different
// 16-byte bank groups, and the recursions' 8-byte reads of 32 consecutive lanes are conflict-free (see "LDS layout" above).
// (An entry-major row [6][GW] was measured in round 2: six 8-byte staging stores per item instead of three 16-byte ones cost
// more than the conflicts they removed; the two-table form keeps the 16-byte stores.)
// PRE (turbo_decode): the channel factors do not change between the passes of a decode, so they are evaluated ONCE per
// launch (signed_q below) and a pass reads copysign(Qa, r0), copysign(Qb, r1) where it would read r0, r1: one exp per item
// and pass -- the prior -- instead of three.
__device__ __forceinline__ double signed_q(double r, double k4) { return __builtin_copysign(exp(k4 * fabs(r)), r); }
// P(bit = 0) from the a-priori LLR, as the reference writes it (turbo.py:239)
__device__ __forceinline__ double prior0(double L) { return 1.0 / (1.0 + exp(L)); }

template <int LGS, bool PRE, bool LIT = false, bool S32 = false, bool TM = false>
__device__ __forceinline__ void stage_chunk(const Ctx<LGS> &c, const RawChunk &rc, double nv2, int ncw, bool llr_in = false) {
    const int GW = c.GW;
    const double k4 = -4.0 / nv2;
    const double lim = T_A * nv2;
#pragma unroll
    for (int q = 0; q < 2; q++) {
        int gg, tl;
        item_map<TM>(c.lane, q, gg, tl);
        if (gg < GW) {
            const double r0 = rc.r0[q], r1 = rc.r1[q], li = rc.li[q];
            if (LIT) {
                // the LITERAL kernel (map_literal_kernel): _compute_branch_prob as written (:62-76) -- four absolute probabilities,
                // underflow included -- and the priors of :239-240
                double g[4];
#pragma unroll
                for (int cc = 0; cc < 4; cc++) {
                    const double x = r0 - (double)(2 * (cc >> 1) - 1);   // code_symbol = 2 * code_bit - 1 (:67-71)
                    const double y = r1 - (double)(2 * (cc & 1) - 1);
                    g[cc] = exp(-(x * x + y * y) / nv2);                 // (:74); nv2 = 2 * noise_variance
                }
                const double p0 = prior0(li);
                double2 *row = reinterpret_cast<double2 *>(c.gam + tl * c.GS + gg * 4);
                row[0] = make_double2(g[0], g[1]);
                row[1] = make_double2(g[2], g[3]);
                *reinterpret_cast<double2 *>(c.pri + tl * c.PS + gg * 2) = make_double2(p0, 1.0 - p0);
                continue;
            }
            const double qa = PRE ? fabs(r0) : exp(k4 * fabs(r0)), qb = PRE ? fabs(r1) : exp(k4 * fabs(r1));
            if (!PRE) {                                           // (A); turbo_decode checks the received values once, at its start
                const double u0 = fabs(r0) + 1.0, u1 = fabs(r1) + 1.0;
                flag_or(q ? c.bad_i1 : c.bad_i0, !(u0 * u0 + u1 * u1 <= lim));
            }
            // priors exactly as the reference forms them (:239-240): e^L may overflow (p0 = 0), 1 - p0 may cancel to 0.
            // PRE (turbo_decode): the stage kernel that produced L_int has already evaluated prior0(L_int) -- the same two operations,
            // in a kernel that waits for memory -- and the slab holds THAT; a pass never needs L_int itself (it writes L - L_int).
            // That is an exp and a division per item and staging, a quarter of a pass's instructions.  (A codeword slot past the end
            // of the batch reads zeros: it gets the prior of L_int = 0.)
            double p0 = PRE ? (gg < ncw ? li : 0.5) : prior0(li), p1 = 1.0 - p0;
            if (PRE && S32) dec_prior(gg < ncw ? li : 0.5, p0, p1);   // float32 slab: the smaller of (p0, p1), signed
            if (TM && llr_in) {                                       // wave-uniform: the last MAP 2 of a decode reads L_int_2 itself (it
                p0 = gg < ncw ? prior0(li) : 0.5;                     // needs it for the decision): the prior as the reference forms it
                p1 = 1.0 - p0;
            }
            // sign of the received value; the sign BIT, so that an underflowed factor stored as -0.0 keeps its sign
            const bool n0 = PRE ? __double2hiint(r0) < 0 : r0 < 0.0, n1 = PRE ? __double2hiint(r1) < 0 : r1 < 0.0;
            const double a0 = n0 ? 1.0 : qa, a1 = n0 ? qa : 1.0;  // systematic bit 0 (c0 = -1) / 1 (c0 = +1)
            const double b0 = n1 ? 1.0 : qb, b1 = n1 ? qb : 1.0;  // parity bit
            double2 *row = reinterpret_cast<double2 *>(c.gam + tl * c.GS + gg * 4);
            row[0] = make_double2(a0 * b0, a0 * b1);              // code = 2*sys_bit + parity_bit
            row[1] = make_double2(a1 * b0, a1 * b1);
            *reinterpret_cast<double2 *>(c.pri + tl * c.PS + gg * 2) = make_double2(p0, p1);
        }
    }
    asm volatile("" ::: "memory");
}

// ---- serial recursions over one staged chunk -----------------------------------------------------------------------
// The operands of a step that come from LDS (branch probabilities and prior weights of this lane's two branches) do not
// depend on the recursion, so a full chunk first issues ALL its LDS reads and forms the branch weights w = gamma * prior
// off the chain; the chain itself is then, per step, one cross-lane exchange, a multiply and a fused multiply-add on
// registers.  (Reading them step by step put an LDS round trip -- s_waitcnt lgkmcnt(0) -- on every step: rocprofv3 showed
// 55 % of the wave cycles waiting at 45 % VALU utilisation.)  The sum-normalisation of (:110-111, :155-158) is applied every
// KNORM steps with the hardware reciprocal: any positive common factor is a valid normalisation.

// branch weights of this lane for the beta recursion at step tl: "lo"/"hi" are the branches into successor s>>1 /
// 2|(s>>1) on the shift-register fast path, the branches of input 0 / 1 otherwise
// (LIT: w_lo / w_hi are the bare priors -- the literal kernel multiplies (beta * gamma) * prior in the reference's order)
template <int LGS, bool SR, bool LIT = false>
__device__ __forceinline__ void beta_w(const Ctx<LGS> &c, int tl, double &g_lo, double &g_hi, double &w_lo, double &w_hi) {
    const double *it = c.gam + tl * c.GS + c.g * 4, *pr = c.pri + tl * c.PS + c.g * 2;
    if (LGS == 2 && SR) {
        g_lo = it[c.o_glo]; g_hi = it[c.o_ghi];
        w_lo = LIT ? pr[c.ilo] : g_lo * pr[c.ilo]; w_hi = LIT ? pr[1 - c.ilo] : g_hi * pr[1 - c.ilo];
    } else {
        g_lo = it[c.code[0]]; g_hi = it[c.code[1]];
        w_lo = LIT ? pr[0] : g_lo * pr[0]; w_hi = LIT ? pr[1] : g_hi * pr[1];
    }
}

// weights of the two incoming branches of this lane's state, in np.where order (:136-138)
template <int LGS>
__device__ __forceinline__ void alpha_w(const Ctx<LGS> &c, int tl, double &w0, double &w1) {
    const double *it = c.gam + tl * c.GS + c.g * 4, *pr = c.pri + tl * c.PS + c.g * 2;
    w0 = it[c.pcode[0]] * pr[c.pin[0]];
    w1 = it[c.pcode[1]] * pr[c.pin[1]];
}
// literal kernel: gamma and prior of the two incoming branches, separately
template <int LGS>
__device__ __forceinline__ void alpha_gp(const Ctx<LGS> &c, int tl, double &g0, double &g1, double &p0, double &p1) {
    const double *it = c.gam + tl * c.GS + c.g * 4, *pr = c.pri + tl * c.PS + c.g * 2;
    g0 = it[c.pcode[0]]; p0 = pr[c.pin[0]];
    g1 = it[c.pcode[1]]; p1 = pr[c.pin[1]];
}

// beta of this lane's two successors (see beta_w for the order)
template <int LGS, bool SR>
__device__ __forceinline__ void beta_nbrs(const Ctx<LGS> &c, double b, double &lo, double &hi) {
    if (LGS == 2 && SR) {
        lo = dppd<0x50>(b);                                       // quad_perm [0,0,1,1]: lane s <- lane s>>1
        hi = dppd<0xFA>(b);                                       // quad_perm [2,2,3,3]: lane s <- lane 2|(s>>1)
    } else {
        exchange2<LGS>(c, b, c.nxt[0], c.nxt[1], lo, hi);
    }
}

// One beta step (:106-108): b <- sum_i b[next(s,i)] * gamma(code(s,i)) * prior(i).  X: the branch products
// x_i = a_own * gamma_i * b[next(s,i)] of (:141-143) are parked at slot i of c.xs (a_own = alpha of this lane's state
// at the step, b = beta of the step's upper time before the update).
template <int LGS, bool SR, bool X, bool LIT = false>
__device__ __forceinline__ void beta_step(const Ctx<LGS> &c, int tl, double &b, double a_own, double g_lo, double g_hi,
                                          double w_lo, double w_hi) {
    double lo, hi;
    beta_nbrs<LGS, SR>(c, b, lo, hi);
    if (X) {
        double *xo = c.xs + tl * XS_ROW + c.lane;                 // idle lanes own a slot too
        const int s_lo = (LGS == 2 && SR) ? c.ilo : 0;            // the input of the "lo" branch: its product goes to that input's row
        xo[s_lo * 64] = (a_own * g_lo) * lo;
        xo[(1 - s_lo) * 64] = (a_own * g_hi) * hi;
    }
    if (LIT) {
        // (:106-111) literally: acc = (b[next] * gamma) * prior summed over the two inputs (two terms: the order of the addition does
        // not matter), then the column divided by its NumPy-ordered sum -- EVERY step, IEEE division, 0 / 0 = NaN as in the reference
        b = (lo * g_lo) * w_lo + (hi * g_hi) * w_hi;
        b = b / np_group_sum<LGS>(b);
        return;
    }
    b = __builtin_fma(hi, w_hi, lo * w_lo);
    if ((tl & (KNORM - 1)) == 0) {
        const double sum = group_sum<LGS>(b);
        flag_or(c.bad_s, c.active && !(sum >= T_SMALL));          // (C)
        b = b * __builtin_amdgcn_rcp(sum);
    }
}

// literal kernel (:136-138, :155-158): (alpha[pred] * gamma) * prior over the two incoming branches, column / its NumPy-ordered sum
template <int LGS, bool SR>
__device__ __forceinline__ void alpha_step_lit(const Ctx<LGS> &c, double &a, double g0, double g1, double p0, double p1) {
    double ap0, ap1;
    exchange_pred<LGS, SR>(c, a, ap0, ap1);
    a = (ap0 * g0) * p0 + (ap1 * g1) * p1;
    a = a / np_group_sum<LGS>(a);
}

template <int LGS, bool SR>
__device__ __forceinline__ void alpha_step(const Ctx<LGS> &c, int tl, double &a, double w0, double w1) {
    double ap0, ap1;
    exchange_pred<LGS, SR>(c, a, ap0, ap1);
    a = __builtin_fma(ap1, w1, ap0 * w0);
    if ((tl & (KNORM - 1)) == KNORM - 1) {
        const double sum = group_sum<LGS>(a);
        flag_or(c.bad_s, c.active && !(sum >= T_SMALL));          // (C)
        a = a * __builtin_amdgcn_rcp(sum);
    }
}

// beta over the `len` staged steps, downwards.  X: also parks the branch products; arow[tl] = alpha of this lane's state
// at step tl (full chunks: registers; the partial last chunk of a block runs rolled and keeps them in c.rw)
template <int LGS, bool SR, bool X, bool LIT = false>
__device__ __forceinline__ void beta_chunk(const Ctx<LGS> &c, double &b, int len, const double (&arow)[CH]) {
    if (len == CH) {
        double gl[CH], gh[CH], wl[CH], wh[CH];
#pragma unroll
        for (int tl = 0; tl < CH; tl++) beta_w<LGS, SR, LIT>(c, tl, gl[tl], gh[tl], wl[tl], wh[tl]);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int tl = CH - 1; tl >= 0; --tl) beta_step<LGS, SR, X, LIT>(c, tl, b, arow[tl], gl[tl], gh[tl], wl[tl], wh[tl]);
    } else {
#pragma unroll 1
        for (int tl = len - 1; tl >= 0; --tl) {
            double gl, gh, wl, wh;
            beta_w<LGS, SR, LIT>(c, tl, gl, gh, wl, wh);
            beta_step<LGS, SR, X, LIT>(c, tl, b, X ? c.rw[tl * 64 + c.lane] : 0.0, gl, gh, wl, wh);
        }
    }
    asm volatile("" ::: "memory");
}

// alpha over the `len` staged steps, upwards; KEEP: arow[tl] (c.rw for a partial chunk) = alpha before step tl
template <int LGS, bool SR, bool KEEP, bool LIT = false>
__device__ __forceinline__ void alpha_chunk(const Ctx<LGS> &c, double &a, int len, double (&arow)[CH]) {
    if (LIT) {                                                    // the literal kernel: LDS operands read step by step
        if (len == CH) {
#pragma unroll
            for (int tl = 0; tl < CH; tl++) {
                double g0, g1, p0, p1;
                alpha_gp<LGS>(c, tl, g0, g1, p0, p1);
                if (KEEP) arow[tl] = a;
                alpha_step_lit<LGS, SR>(c, a, g0, g1, p0, p1);
            }
        } else {
#pragma unroll 1
            for (int tl = 0; tl < len; tl++) {
                double g0, g1, p0, p1;
                alpha_gp<LGS>(c, tl, g0, g1, p0, p1);
                if (KEEP) c.rw[tl * 64 + c.lane] = a;
                alpha_step_lit<LGS, SR>(c, a, g0, g1, p0, p1);
            }
        }
        asm volatile("" ::: "memory");
        return;
    }
    if (len == CH) {
        double w0[CH], w1[CH];
#pragma unroll
        for (int tl = 0; tl < CH; tl++) alpha_w<LGS>(c, tl, w0[tl], w1[tl]);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int tl = 0; tl < CH; tl++) {
            if (KEEP) arow[tl] = a;
            alpha_step<LGS, SR>(c, tl, a, w0[tl], w1[tl]);
        }
    } else {
#pragma unroll 1
        for (int tl = 0; tl < len; tl++) {
            double w0, w1;
            alpha_w<LGS>(c, tl, w0, w1);
            if (KEEP) c.rw[tl * 64 + c.lane] = a;
            alpha_step<LGS, SR>(c, tl, a, w0, w1);
        }
        asm volatile("" ::: "memory");
    }
}

// time-parallel epilogue of a phase-2 chunk: app sums in state order, L = L_int + log(app1/app0) (:145)
template <int LGS, bool BITS, bool LIT = false, bool S32 = false, bool TM = false>
__device__ __forceinline__ void epilogue(const Ctx<LGS> &c, const PassIO &io, const double (&li)[2], int t_lo, int len) {
    constexpr int S = Ctx<LGS>::S;
    asm volatile("" ::: "memory");
#pragma unroll
    for (int q = 0; q < 2; q++) {
        int gg, tl;
        item_map<TM>(c.lane, q, gg, tl);
        const unsigned m = item_mask(gg, io.ncw, tl, len);        // no such item: computes on whatever LDS holds, stores nowhere
        const bool ok = m != 0;
        // the S products of input 0 and of input 1 of this item: the lanes of codeword slot gg are gg * S .. gg * S + S - 1
        const double2 *x0 = reinterpret_cast<const double2 *>(c.xs + tl * XS_ROW + (gg < c.GW ? gg : 0) * S), *x1 = x0 + 32;
        double app0 = 0.0, app1 = 0.0;
#pragma unroll
        for (int st = 0; st < S / 2; st++) {                      // sums in state order (:141-143)
            const double2 v0 = x0[st], v1 = x1[st];
            app0 += v0.x; app0 += v0.y;
            app1 += v1.x; app1 += v1.y;
        }
        // turbo_decode, every pass whose output only ever becomes the NEXT pass's prior (all but the last two, round 4): the stage
        // kernel would turn E = log(app1 / app0) into prior0(E) = 1 / (1 + e^E) = app0 / (app0 + app1) -- the logarithm here and the
        // exponential there cancel, so the pass writes that quotient itself and the stage kernel only permutes.  A second division
        // instead of the log (45 instructions per item: ablation 4.53 -> 4.23 ms per config-3 decode with all twelve logs gone); the
        // priors differ from the reference's log -> exp round trip by its own rounding (~|E| 1e-16 relative).
        double L;
        if (LIT) {
            L = li[q] + log(app1 / app0);                         // (:145) as written: 0 / 0, log 0 and log inf give the reference's NaN / -+inf
        } else if (io.pout) {
            // ... as 1 / (1 + r), r = app1 / app0 -- the reference's own last two operations with r in the place of e^{log r}: the exact
            // zeros of its priors (1 + e rounds to 1 below e = 2^-53: p1 = 0; e = inf: p0 = 0) fall where the reference has them, which
            // app0 / (app0 + app1) does not guarantee (caught by the extreme-regime fixtures)
            const double r = app1 / app0;
            L = S32 ? enc_prior_odds(r) : 1.0 / (1.0 + r);
            flag_or(q ? c.bad_i1 : c.bad_i0, ok && (!(r > 0.0 && r < __builtin_huge_val()) || !(fmax(app0, app1) >= T_SMALL)));   // (D): log r not finite; (E)
        } else {
#ifdef CPX_AB_NO_EPILOG_LOG                                        /* ablation builds only (experiments/README.md): what the logarithm costs */
            const double lr = app1 / app0 - 1.0;
#else
            const double lr = fast_log(app1 / app0);
#endif
            flag_or(q ? c.bad_i1 : c.bad_i0, ok && (!(fabs(lr) < __builtin_huge_val()) || !(fmax(app0, app1) >= T_SMALL)));   // (D), (E)
            L = io.ext ? lr : li[q] + lr;
        }
        const unsigned t = (unsigned)(t_lo + tl);
        if (TM) {
            // row t of the pass's output array: eight lanes write one 64-byte half of a 128-byte row, a chunk eight consecutive rows.
            // The last MAP 2 stores its decision L_2 = L_int_2 + E_2 > 0 (:148-152, :326) INSTEAD -- one byte per slot, rows of TM_RW
            // bytes at the start of the same array (nobody reads its E_2) -- for turbo_final_kernel to de-interleave (:331).
            const unsigned ms = item_mask(gg, c.GW, tl, len);         // (slots past the end of the batch are the pair's own: written too)
            const unsigned e = t * TM_RW + (unsigned)gg;
            slab_st<S32>(io.rsys, ms & io.st_out, e, io.oout, L);
            __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(li[q] + L > 0 ? 1 : 0), io.rsys, item_off(ms & io.st_dec, e), io.odec, 0);
            continue;
        }
        slab_st<S32>(io.rout, m, (unsigned)(gg * io.lstride) + t, io.oout, L);
        if (BITS) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)((io.want_bits && L > 0) ? 1 : 0), io.rbits,
                                             item_off(m, (unsigned)(gg * io.N) + t), 0, 0);               // (:148-152)
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
template <int LGS, bool SR, bool PRE, bool LIT = false, bool S32 = false, bool TM = false>
__device__ void map_pass(const Ctx<LGS> &c, const PassIO &io) {
    const int N = io.N;
    const int K = (N + CH - 1) / CH, K1 = K / 2;                  // F: chunks [0,K1) then [K1,K); R: [K1,K) then [0,K1)
    auto clen = [&](int k) { const int r = N - k * CH; return r < CH ? r : CH; };
    // checkpoint row k (state vector at time k*CH), this lane's entry
    auto ck_ld = [&](int k) { return ld_off(io.ckpt, (unsigned)(k * 64 + c.lane)); };
    auto ck_st = [&](int k, double v) { st_off(io.ckpt, (unsigned)(k * 64 + c.lane), v); };
    double arow[CH];                                              // alpha of this lane's state at the chunk's steps
#pragma unroll
    for (int tl = 0; tl < CH; tl++) arow[tl] = 0.0;
    // Either wave walks ALL K chunks in one direction -- F: 0 .. K-1, R: K-1 .. 0 -- first on its own recursion only (phase 1),
    // then on both (phase 2); `seq(i)` is the i-th chunk of the walk.  The received values of a chunk are requested TWO chunks
    // ahead: two register sets X / Y alternate (the loops below are unrolled by two, so that no set is ever copied -- a copy would
    // wait for the load it copies).  Measured on config 3, same box (experiments/README.md): one set 4.90 ms, two sets 4.85 ms,
    // two sets + counted waits (PassIO) 4.80 ms -- the pass is NOT limited by its loads: rocprofv3 has the two waves of a SIMD
    // issuing VALU 53 % of the time, 22 % of the wave cycles in s_waitcnt and 29 % ready-but-not-issued.
    auto seq = [&](int i) { return c.fwd ? i : K - 1 - i; };
    // (past the end of the walk the last chunk is requested again: an `if` around the loads would bring the uncounted wait back)
    auto fetch = [&](RawChunk &S, int i) {
        const int k = seq(i < K ? i : K - 1), kn = seq(i + 2 < K ? i + 2 : K - 1);     // kn: what this set is loaded with next (TM)
        load_raw<LGS, S32, TM>(c, io, S, k * CH, clen(k), kn * CH);
    };
    RawChunk X, Y;
    if (TM) {                                                     // row indices of the first two chunks of the walk
        const int ta = seq(0) * CH + (c.lane >> 3), tb = seq(K > 1 ? 1 : 0) * CH + (c.lane >> 3);
        X.ridx = (unsigned)io.idx[ta < N ? ta : N - 1];
        Y.ridx = (unsigned)io.idx[tb < N ? tb : N - 1];
    }
    fetch(X, 0);
    fetch(Y, 1);
    // runs step(i, set) for i = i0 .. i1-1, alternating X, Y; an odd count ends with the sets swapped by value (once per phase)
    // ... and stops (returns true) as soon as this wave has raised a "detect and redo" flag where the redo covers the whole pair
    // (io.abort_ok): above Es/N0 ~ 14 dB flag (A) fires in the first chunk of every pair and the fast pass costs next to nothing
    // (8.6 us for a fully flagged config-3 batch, profiles/r05_map_highsnr_pmc.json)
    auto dead = [&]() { return !LIT && io.abort_ok && (c.bad_s | c.bad_i0 | c.bad_i1) != 0; };
    auto run = [&](int i0, int i1, auto &&step) {
        int i = i0;
        for (; i + 1 < i1; i += 2) {
            step(i, X);
            step(i + 1, Y);
            if (dead()) return true;
        }
        if (i < i1) {
            step(i, X);
            const RawChunk t = X; X = Y; Y = t;
        }
        return dead();
    };
    // Phase 2 of either wave, one chunk: alpha over the chunk (kept per step), then beta over it with the branch products,
    // then the epilogue.  One of the two recursions continues the wave's own chain, the other starts from the partner's
    // checkpoint and is discarded afterwards.  The checkpoint is loaded at the TOP of the iteration, before the prefetch
    // of a later chunk is issued: loads return in order, so the chain then only waits for that one load.
    // The epilogue of a chunk runs at the top of the NEXT iteration, after that iteration's checkpoint load has been
    // issued: loads and stores retire through one in-order counter, so with the LLR stores issued BEFORE the
    // checkpoint load the chain waited for their write acknowledgements as well.
    double li_prev[2] = {0.0, 0.0};
    int t_prev = 0;
    int len_prev = 0;                                             // first iteration of phase 2: nothing to write
    if (c.fwd) {
        // ---------------- phase 1: alpha over chunks 0 .. K1-1, checkpoint before every chunk ----------------
        double a = (c.s == 0) ? 1.0 : 0.0;                        // f_state_metrics[0][0] = 1 (:221)
        run(0, K1, [&](int i, RawChunk &S) {
            ck_st(i, a);                                          // alpha at time i*CH (read by R in phase 2)
            stage_chunk<LGS, PRE, LIT, S32, TM>(c, S, io.nv2, io.ncw, io.llr_in);
            fetch(S, i + 2);
            alpha_chunk<LGS, SR, false, LIT>(c, a, CH, arow);          // chunks below K1 are full
        });
        pair_sync();                                              // (a stopped wave still meets the workgroup here)
        // ---------------- phase 2: chunks K1 .. K-1: own alpha, beta from R's checkpoint, combine ----------------
        if (!dead() && !run(K1, K, [&](int k, RawChunk &S) {
            const int len = clen(k);
            double b = ck_ld(k + 1);                              // beta at the upper boundary of chunk k
            epilogue<LGS, !PRE, LIT, S32, TM>(c, io, li_prev, t_prev, len_prev);
            stage_chunk<LGS, PRE, LIT, S32, TM>(c, S, io.nv2, io.ncw, io.llr_in);
            li_prev[0] = S.li[0]; li_prev[1] = S.li[1];
            fetch(S, k + 2);
            alpha_chunk<LGS, SR, true, LIT>(c, a, len, arow);
            beta_chunk<LGS, SR, true, LIT>(c, b, len, arow);
            t_prev = k * CH; len_prev = len;
        }))
            epilogue<LGS, !PRE, LIT, S32, TM>(c, io, li_prev, t_prev, len_prev);
    } else {
        // ---------------- phase 1: beta over chunks K-1 .. K1, checkpoint before every chunk ----------------
        double b = 1.0;                                           // b_state_metrics[:, N] = 1 (:225)
        run(0, K - K1, [&](int i, RawChunk &S) {
            const int k = K - 1 - i;
            ck_st(k + 1, b);                                      // beta at the upper boundary of chunk k (read by F)
            stage_chunk<LGS, PRE, LIT, S32, TM>(c, S, io.nv2, io.ncw, io.llr_in);
            fetch(S, i + 2);
            beta_chunk<LGS, SR, false, LIT>(c, b, clen(k), arow);
        });
        pair_sync();
        // ---------------- phase 2: chunks K1-1 .. 0 (all full): alpha from F's checkpoint, own beta, combine ----------------
        if (!dead() && !run(K - K1, K, [&](int i, RawChunk &S) {
            const int k = K - 1 - i;
            double a = ck_ld(k);                                  // alpha at the lower boundary of chunk k
            epilogue<LGS, !PRE, LIT, S32, TM>(c, io, li_prev, t_prev, len_prev);      // of the previous chunk (see the forward wave)
            stage_chunk<LGS, PRE, LIT, S32, TM>(c, S, io.nv2, io.ncw, io.llr_in);
            li_prev[0] = S.li[0]; li_prev[1] = S.li[1];
            fetch(S, i + 2);
            alpha_chunk<LGS, SR, true, LIT>(c, a, CH, arow);           // arow[tl] = alpha at time k*CH + tl
            beta_chunk<LGS, SR, true, LIT>(c, b, CH, arow);
            t_prev = k * CH; len_prev = CH;
        }))
            epilogue<LGS, !PRE, LIT, S32, TM>(c, io, li_prev, t_prev, len_prev);
    }
    if (!LIT) publish_flags<LGS, TM>(c, io.flags, io.ncw);
}

struct MapParams {
    MapTables tb;
    const double *sys, *par, *Lin;     // [B][N]
    double *Lout;                      // [B][N]
    uint8_t *bits;                     // [B][N]
    double *scratch;                   // per pair: checkpoint rows [nchunks + 1][64]
    uint8_t *flags;                    // [B] "detect and redo" (zeroed before the launch), may be null
    int64_t B, N;
    double nv2;
    int want_bits, GW;
};

template <int LGS, bool SR>
__global__ __launch_bounds__(128 * NPAIR) void map_decode_kernel(MapParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Ctx<LGS> c;
    init_ctx<LGS>(c, p.tb, smem, p.GW);
    // wave-uniform by construction; readfirstlane tells the compiler, so that everything derived from it (the pair's base
    // pointers, codeword counts) lives in SGPRs and the loads take the `saddr + 32-bit voffset` form
    const int64_t pair = (int64_t)blockIdx.x * (blockDim.x >> 7) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 7));
    const int64_t K = (p.N + CH - 1) / CH;
    const int64_t cw0 = pair * p.GW, o0 = cw0 * p.N;
    const int64_t left = p.B - cw0;
    PassIO io;
    io.N = (int)p.N; io.sstride = io.pstride = io.lstride = io.N; io.ext = false; io.pout = false; io.llr_in = false; io.sysg = 0u;
    io.ncw = (int)(left < p.GW ? left : p.GW); io.nv2 = p.nv2;
    const unsigned span = io.ncw > 0 ? OOB : 0u;                  // a pair past the end of the batch: everything out of range
    io.rsys = pass_buffer(p.sys + o0, span); io.rpar = pass_buffer(p.par + o0, span);
    io.rlin = pass_buffer(p.Lin + o0, span); io.rout = pass_buffer(p.Lout + o0, span);
    io.rbits = pass_buffer(p.bits ? p.bits + o0 : nullptr, p.bits ? span : 0u);
    io.osys = io.opar = io.olin = io.oout = 0u;
    io.want_bits = p.want_bits;
    io.ckpt = p.scratch + pair * (K + 1) * 64;
    io.flags = p.flags ? p.flags + cw0 : nullptr;
    io.abort_ok = io.flags != nullptr;
    map_pass<LGS, SR, false>(c, io);                              // L_ext and the hard decisions leave in the pass's epilogue
}

// ---- the LITERAL wave-parallel kernel: the redo path behind map_decode_kernel (round 5) ------------------------------------------
// Round 3's redo path (bcjr_exact.hip, one codeword per LANE, state vectors in HBM scratch) is literal but serial: 7 ms per flagged
// codeword, 28 ms when flag (A) sends a whole batch there -- which it does above Es/N0 ~ 14 dB (sigma^2 = 0.01: 0.35 -> 28 ms, a silent
// 80 x cliff).  This is the same reference arithmetic -- absolute gamma, priors 1 / (1 + e^L) and 1 - p0, (metric * gamma) * prior
// summed over the two branches, every column divided by its NumPy-ordered sum at every step, app sums in state order, L_int +
// log(app1 / app0) -- on the wave-pair mapping of map_decode_kernel: one state per lane, forward and reverse wave, checkpoints, the
// recomputed chain "by the same operations", so every value is the one the sequential reference loop produces, underflow, 0 / 0 and
// log 0 included (tests/test_abnormal_golden_gpu.py).  A pair none of whose codewords is flagged only joins the workgroup's one
// barrier and leaves; a pair with a flagged codeword decodes all GW of them again (literal results for the unflagged ones are as valid
// as the fast ones).  `redo` counts the flagged codewords for cpx_last_kernel's string in a DEVICE word that the launch's last
// workgroup publishes (cpx_internal.h redo_finish; the first version incremented a pinned host word from every pair: 1024 atomics
// over PCIe cost ~1 ms of a 1.5 ms launch, whatever the arithmetic did).
template <int LGS, bool SR>
__global__ __launch_bounds__(128 * NPAIR) void map_literal_kernel(MapParams p, RedoCounter redo) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int64_t pair = (int64_t)blockIdx.x * (blockDim.x >> 7) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 7));
    const int64_t cw0 = pair * p.GW, left = p.B - cw0;
    const int ncw = (int)(left < p.GW ? left : p.GW);
    const int lane = threadIdx.x & 63;
    const unsigned long long hit = __ballot(lane < ncw && p.flags[cw0 + (lane < ncw ? lane : 0)] != 0);
    __shared__ unsigned wg_redo;
    if (threadIdx.x == 0) wg_redo = 0;
    __syncthreads();
    if ((threadIdx.x & 127) == 0 && hit != 0) atomicAdd(&wg_redo, (unsigned)__popcll(hit));
    if (hit == 0) {                                               // pair-uniform: both waves of the pair read the same bytes
        pair_sync();                                              // the other pairs of the workgroup meet here once (map_pass)
        redo_finish(redo, wg_redo);
        return;
    }
    Ctx<LGS> c;
    init_ctx<LGS>(c, p.tb, smem, p.GW);
    const int64_t K = (p.N + CH - 1) / CH, o0 = cw0 * p.N;
    PassIO io;
    io.N = (int)p.N; io.sstride = io.pstride = io.lstride = io.N; io.ext = false; io.pout = false; io.llr_in = false; io.sysg = 0u;
    io.ncw = ncw; io.nv2 = p.nv2;
    io.rsys = pass_buffer(p.sys + o0, OOB); io.rpar = pass_buffer(p.par + o0, OOB);
    io.rlin = pass_buffer(p.Lin + o0, OOB); io.rout = pass_buffer(p.Lout + o0, OOB);
    io.rbits = pass_buffer(p.bits ? p.bits + o0 : nullptr, p.bits ? OOB : 0u);
    io.osys = io.opar = io.olin = io.oout = 0u;
    io.want_bits = p.want_bits;
    io.ckpt = p.scratch + pair * (K + 1) * 64;
    io.flags = nullptr; io.abort_ok = false;
    map_pass<LGS, SR, false, true>(c, io);
    redo_finish(redo, wg_redo);
}

struct TurboParams {
    MapTables tb;
    const double *sys, *p1, *p2, *Lint;   // [B][N], Lint may be null
    const int32_t *perm;                  // [N]
    const int32_t *iperm, *ident;         // [N] each: the inverse permutation and the identity (turbo_tables_kernel)
    uint8_t *bits;                        // [B][N]
    double *ckpt;                         // per pair: checkpoint rows [nchunks + 1][64]
    // The slab (round 6: TIME-MAJOR).  Per ROW GROUP five arrays of N rows; a row = RW = 16 codeword slots at one time step = one
    // 128-byte line.  A pair of wavefronts decodes GW codewords, so a row group is 16 / GW consecutive pairs (one pair for the
    // 16 codewords of a 4-state trellis, two for 8 states, ...) -- a gathered row is a whole line whatever the trellis.  Arrays:
    //   0  what MAP 1 writes and MAP 2 reads as its prior     1  what MAP 2 writes and MAP 1 reads (initially prior0(L_int_1))
    //   2, 3, 4  the signed channel factors (signed_q) of sys_symbols, non_sys_symbols_1, non_sys_symbols_2
    // all in the NATURAL order of the pass that wrote them.  The interleaver is the same for every codeword
    // (interleavers.py:13-47), so interlv / deinterlv (:310, :319, :329) move whole ROWS: a pass reads the row perm[t] (MAP 2; its
    // systematic factors too) or inverse_perm[t] (MAP 1) of the other pass's output where it used to read row t of a permuted copy --
    // eight 128-byte lines per load instruction either way -- and the thirteen permutation launches of rounds 3-5 (4.4 of a decode's
    // 17.9 GB, 0.9 of its 3.9 ms) are gone.  (History: rows [B][7][N] until round 5 -- folding the permutation into THAT layout made
    // every permuted 8-byte access its own memory request: 5.15 / 6.65 ms against 3.9; a chunked slab [pair][array][chunk][GW][8] was
    // tried in round 3 and is codeword-major inside a chunk, so it could not fold either.  experiments/README.md.)
    double *larr;
    // (the decisions of the last MAP 2 -- [N][RW] bytes, in ITS order -- take the place of its output at the start of array 1)
    uint8_t *flags;                       // [B] "detect and redo" (zeroed before the launch), may be null
    int64_t B, N;
    double nv2;
    int n_iter, GW, RW;
    int lgG;                              // pairs per row group = 16 / GW = 1 << lgG: codeword slot g of group r is codeword 16 r + g
    // Every pair of wavefronts of the chip walks its arrays at the same pace: array and group strides are kept away from multiples
    // of a large power of two -- an array has NR = N | 1 rows, an odd number of 128-byte lines, so row t of group r sits r * odd lines
    // away from group 0's: all residues of any channel interleave.
    int64_t NR;
};

// ---- turbo_decode as a SEQUENCE of launches (round 3) ------------------------------------------------------------------------
// One launch per MAP pass (the pass of map_decode_kernel, reading the time-major slab) between turbo_init_kernel and
// turbo_final_kernel.  Round 2 ran everything in ONE persistent launch; inlined into that kernel's iteration loop the pass needed
// more than 256 VGPRs -- 18 to 23 of them spilled, and a scratch reload retires through the same in-order counter as the prefetches,
// i.e. it waits for the HBM round trip issued just before it (rocprofv3, round 2: 61 % of the wave cycles parked in s_waitcnt
// where the stand-alone pass has 33 %).  As its own kernel the pass keeps the register allocation of map_decode_kernel (no
// scratch); a kernel boundary costs ~2 us.
//   `second`: MAP 2 of an iteration (:326) else MAP 1 (:315);  `idx`: the row index table of the pass's prior (see TurboParams);
//   `pout`: write prior0(E) instead of E (epilogue);  `llr_in`: the prior array holds LLRs (the pass before wrote E itself);
//   `last`: the final MAP 2 -- it stores decisions and nothing else.
template <int LGS, bool SR, bool S32 = false>
__global__ __launch_bounds__(128 * NPAIR) void turbo_pass_kernel(TurboParams p, const int32_t *idx, int second, int pout, int llr_in, int last) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Ctx<LGS> c;
    init_ctx<LGS>(c, p.tb, smem, p.GW);
    const int64_t pair = (int64_t)blockIdx.x * (blockDim.x >> 7) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 7));
    const int64_t N = p.N, K = (N + CH - 1) / CH, cw0 = pair * p.GW;
    const int64_t left = p.B - cw0;
    PassIO io;
    io.lstride = io.sstride = io.pstride = 0;                     // (codeword-major strides: unused)
    io.N = (int)N; io.nv2 = p.nv2; io.want_bits = last;
    io.ncw = (int)(left < p.GW ? left : p.GW);
    io.ckpt = p.ckpt + pair * (K + 1) * 64;
    io.flags = p.flags ? p.flags + cw0 : nullptr;
    io.abort_ok = false;                                          // turbo's redo is per codeword: the other codewords of a pair must be finished
    io.ext = true;                                                // the pass writes E = L - L_int (:318, :328) ...
    io.pout = pout != 0;                                          // ... or prior0(E) directly (epilogue)
    io.llr_in = llr_in != 0;
    io.sysg = second ? ~0u : 0u;                                  // MAP 2 decodes interlv(sys_symbols) (:310): same rows as its prior
    constexpr unsigned ES = S32 ? 4u : 8u;                         // bytes per slab element ("fp32-fast": float32 slab)
    const unsigned nb = (unsigned)p.NR * (unsigned)p.RW * ES;      // bytes of one array of a row group
    const int64_t group = pair >> p.lgG;
    const unsigned so = (unsigned)(pair & ((1 << p.lgG) - 1)) * (unsigned)p.GW;       // first slot of this pair in its group's rows
    const char *base = reinterpret_cast<const char *>(p.larr) + group * 5 * (int64_t)nb;
    const bool live = io.ncw > 0;                                  // a pair past the end of the batch: everything out of range
    io.rsys = io.rpar = io.rlin = io.rout = io.rbits = pass_buffer(base, live ? OOB : 0u);
    io.st_out = last ? 0u : ~0u;
    io.st_dec = last ? ~0u : 0u;
    io.idx = idx;
    //   first  half-iteration: [L_ext_1, _] = map_decode(sys,   non_sys_1, trellis, nv, L_int_1, 'compute')   (:315)
    //   second half-iteration: [L_2, bits]  = map_decode(sys_i, non_sys_2, trellis, nv, L_int_2, mode)        (:326)
    io.osys = __builtin_amdgcn_readfirstlane(2u * nb + so * ES);
    io.opar = __builtin_amdgcn_readfirstlane((second ? 4u : 3u) * nb + so * ES);
    io.olin = __builtin_amdgcn_readfirstlane((second ? 0u : nb) + so * ES);
    io.oout = __builtin_amdgcn_readfirstlane((second ? nb : 0u) + so * ES);
    io.odec = __builtin_amdgcn_readfirstlane(nb + so);             // decisions: bytes, at the start of array 1
    map_pass<LGS, SR, true, false, S32, true>(c, io);
}

// inverse permutation and identity, once per decode (the row index tables of the passes)
__global__ void turbo_tables_kernel(const int32_t *perm, int32_t *iperm, int32_t *ident, int64_t N) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) {
        iperm[perm[i]] = (int32_t)i;
        ident[i] = (int32_t)i;
    }
}

// Slab initialisation (:305-310): the caller's arrays [B][N] -> the time-major arrays 1 (prior0 of L_int_1, :239), 2, 3, 4 (channel
// factors, evaluated ONCE per decode) of every pair, transposed through LDS -- a wavefront reads 64 steps of one codeword (512
// contiguous bytes) and the workgroup writes 64 whole rows (8 KB contiguous per array) -- plus flag (A) of "detect and redo".
// Slots past the end of the batch get zeros.  One workgroup per (row group, 64-step tile).
constexpr int IT = 64;                    // steps per tile
constexpr int IT_ROW = IT + 2;            // doubles per codeword row of the LDS tile (2 GW + t: conflict-free b64 column reads)
template <bool S32>
__global__ __launch_bounds__(256) void turbo_init_kernel(TurboParams p, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *tile = reinterpret_cast<double *>(smem);              // [4 arrays][RW][IT_ROW]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t group = blockIdx.x / ntiles, N = p.N;
    const int64_t t0 = (int64_t)(blockIdx.x % ntiles) * IT, t = t0 + lane;
    const int RW = p.RW;

    const double k4 = -4.0 / p.nv2;
    // (A) of "detect and redo" for every received value on its own: |r| > sqrt(T_A nv2 / 2) - 1 bounds
    // (|r0| + 1)^2 + (|r1| + 1)^2 for any pairing
    const double rmax = sqrt(0.5 * T_A * p.nv2) - 1.0;
    for (int g = wv; g < RW; g += 4) {
        const int64_t cw = group * RW + g;                        // slot g of row group r: codeword 16 r + g
        const bool have = cw < p.B && t < N;
        double v[4] = {0.0, 0.0, 0.0, 0.0};
        bool far = false;
        if (have) {
            const double rs = p.sys[cw * N + t], r1 = p.p1[cw * N + t], r2 = p.p2[cw * N + t];
            far = !(fabs(r1) <= rmax) || !(fabs(r2) <= rmax) || !(fabs(rs) <= rmax);
            const double L = p.Lint ? p.Lint[cw * N + t] : 0.0;
            v[0] = S32 ? enc_prior_odds(exp(L)) : prior0(L);      // what a pass reads as its prior (stage_chunk)
            v[1] = signed_q(rs, k4);
            v[2] = signed_q(r1, k4);
            v[3] = signed_q(r2, k4);
        }
        if (p.flags && __ballot(far) != 0 && lane == 0) p.flags[cw] = 1;
#pragma unroll
        for (int a = 0; a < 4; a++) tile[(a * RW + g) * IT_ROW + lane] = v[a];
    }
    __syncthreads();
    using T = typename std::conditional<S32, float, double>::type;
    const int64_t arr = p.NR * RW;                                // elements of one array of a pair
    T *dst = reinterpret_cast<T *>(p.larr) + group * 5 * arr;
    const int rows = (int)(N - t0 < IT ? N - t0 : IT);
    for (int e = threadIdx.x; e < rows * RW; e += 256) {
        const int r = e / RW, g = e - r * RW;
#pragma unroll
        for (int a = 0; a < 4; a++) dst[(a + 1) * arr + (t0 + r) * RW + g] = (T)tile[(a * RW + g) * IT_ROW + r];
    }
}

// decoded_bits = deinterlv(L_2 > 0) (:148-152, :331): the last MAP 2 left its decisions as rows of RW bytes in its own (interleaved)
// order; output position d of every codeword of the row group is row inverse_perm[d].  One workgroup per (row group, 256 positions).
constexpr int FT = 256;
template <bool S32>
__global__ __launch_bounds__(FT) void turbo_final_kernel(TurboParams p, int ntiles) {
    const int64_t group = blockIdx.x / ntiles, N = p.N;
    const int64_t d = (int64_t)(blockIdx.x % ntiles) * FT + threadIdx.x;
    if (d >= N) return;
    constexpr int64_t ES = S32 ? 4 : 8;
    const uint8_t *row = reinterpret_cast<const uint8_t *>(p.larr) + (group * 5 + 1) * p.NR * p.RW * ES + (int64_t)p.iperm[d] * p.RW;
    for (int g = 0; g < p.RW && group * p.RW + g < p.B; g++) p.bits[(group * p.RW + g) * N + d] = row[g];   // consecutive lanes, consecutive bytes
}

// ---- turbo_decode, the LITERAL redo (round 5): flagged pairs decode their codewords again, all iterations, in ONE launch -----------
// Round 3's redo (bcjr_exact.hip turbo_exact_kernel, one codeword per lane) costs 2 x iterations x 7 ms for ANY number of flagged
// codewords.  A pair's codewords are independent of every other pair's, so here a flagged pair runs the reference's whole loop
// (turbo.py:300-331) by itself: map_pass<LIT> for MAP 1 and MAP 2 -- the literal wave-parallel pass of map_literal_kernel -- and
// between them, elementwise in the pair's own rows of the slab (the fast sequence is done with them), what the reference writes:
//   L_ext_1 = L_ext_1 - L_int_1 (:318), L_int_2 = interlv(L_ext_1) (:319), L_int_1 = deinterlv(L_2 - L_int_2) (:328-329),
//   decoded_bits = deinterlv(L_2 > 0) of the last iteration (:148-152, :331).
// The pair's share of its row group's slab (5 NR GW doubles; the fast sequence is done with it) is reused CODEWORD-major, four arrays [N] per
// codeword: 0 L_int_1, 1 a pass's output (L_int + log(app1 / app0), i.e. L_ext_1 before the subtraction / L_2), 2 L_int_2,
// 3 interlv(sys_symbols) as RAW symbols.  ONE pair per workgroup (a pair that is not flagged leaves at once; the barriers below only
// ever meet the two waves of a pair), a rare path: speed is what the wave-parallel pass gives (~0.5 ms per MAP pass of a full
// config-3 batch against 7 ms per flagged codeword before), not a goal of its own.
template <int LGS, bool SR>
__global__ __launch_bounds__(128) void turbo_literal_kernel(TurboParams p, RedoCounter redo) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int64_t pair = blockIdx.x;
    const int64_t N = p.N, K = (N + CH - 1) / CH, cw0 = pair * p.GW, left = p.B - cw0, ls = 4 * N;
    const int ncw = (int)(left < p.GW ? left : p.GW);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long hit = __ballot(lane < ncw && p.flags[cw0 + (lane < ncw ? lane : 0)] != 0);
    if (hit == 0) {
        redo_finish(redo, 0u);
        return;
    }
    Ctx<LGS> c;
    init_ctx<LGS>(c, p.tb, smem, p.GW);
    double *slab = p.larr + pair * 5 * p.NR * p.GW;        // this pair's share of its row group's region (always float64 here)
    auto at = [&](int a, int g, int64_t t) -> double & { return slab[((int64_t)g * 4 + a) * N + t]; };
    // elementwise work: the pair's codewords alternate between its two waves
    for (int g = wave; g < ncw; g += 2)
        for (int64_t t = lane; t < N; t += 64) {
            at(0, g, t) = p.Lint ? p.Lint[(cw0 + g) * N + t] : 0.0;                   // L_int_1 (:305-308)
            at(3, g, t) = p.sys[(cw0 + g) * N + p.perm[t]];                           // interlv(sys_symbols) (:310)
        }
    pair_sync();
    PassIO io;
    io.N = (int)N; io.nv2 = p.nv2; io.want_bits = 0; io.ncw = ncw;
    io.ckpt = p.ckpt + pair * (K + 1) * 64;
    io.flags = nullptr; io.abort_ok = false; io.ext = false; io.pout = false; io.llr_in = false; io.sysg = 0u;
    io.rbits = pass_buffer(nullptr, 0u);
    io.rlin = io.rout = pass_buffer(slab, OOB);
    io.lstride = (int)ls;
    const unsigned nb = (unsigned)N * 8u;
    io.oout = nb;                                                                     // array 1
    for (int it = 0; it < p.n_iter; it++) {
        // MAP 1 (:315): sys, non_sys_1 straight from the caller's arrays, L_int_1 from array 0
        io.rsys = pass_buffer(p.sys + cw0 * N, OOB); io.osys = 0u; io.sstride = (int)N;
        io.rpar = pass_buffer(p.p1 + cw0 * N, OOB); io.opar = 0u; io.pstride = (int)N;
        io.olin = 0u;
        map_pass<LGS, SR, false, true>(c, io);
        pair_sync();
        for (int g = wave; g < ncw; g += 2)
            for (int64_t t = lane; t < N; t += 64) {
                const int64_t s2 = p.perm[t];
                at(2, g, t) = at(1, g, s2) - at(0, g, s2);                            // L_int_2 = interlv(L_ext_1 - L_int_1) (:318-319)
            }
        pair_sync();
        // MAP 2 (:326): interleaved systematic symbols (array 4), non_sys_2, L_int_2 (array 2)
        io.rsys = pass_buffer(slab, OOB); io.osys = 3u * nb; io.sstride = (int)ls;
        io.rpar = pass_buffer(p.p2 + cw0 * N, OOB); io.opar = 0u; io.pstride = (int)N;
        io.olin = 2u * nb;
        map_pass<LGS, SR, false, true>(c, io);
        pair_sync();
        const bool last = it == p.n_iter - 1;
        for (int g = wave; g < ncw; g += 2)
            for (int64_t t = lane; t < N; t += 64) {
                const int64_t d = p.perm[t];
                const double L2 = at(1, g, t);
                at(0, g, d) = L2 - at(2, g, t);                                       // L_int_1 = deinterlv(L_2 - L_int_2) (:328-329)
                if (last) p.bits[(cw0 + g) * N + d] = (uint8_t)(L2 > 0 ? 1 : 0);      // 'decode' (:148-152), deinterlv (:331)
            }
        pair_sync();
    }
    redo_finish(redo, (unsigned)__popcll(hit));
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
    CPX_REQUIRE(t->S >= 2, CPX_ELIMIT, "map_decode: needs at least 2 states (got %d)", t->S);
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
    CPX_TRACE("cpx_map_decode_batch_dev");
    cpx::IssueGuard issue_guard;
    MapParams p;
    int rc = fill_tables(t, p.tb);
    if (rc) return rc;
    CPX_REQUIRE(B >= 0 && N >= 0, CPX_EINVAL, "map_decode: negative size");
    CPX_REQUIRE(N < (1ll << 24), CPX_ELIMIT, "map_decode: blocks of 2^24 steps or more are not supported (31-bit lane offsets over 16 codewords)");
    if (B == 0 || N == 0) return CPX_OK;
    hipStream_t st = pick_stream(stream);
    if (t->S > 16) {                                              // beyond the wave-pair kernels: the literal absolute-scale kernel alone
        rc = bcjr_exact_map(t, d_sys, d_par, d_L_int, B, N, 2 * noise_variance, want_bits, d_L_ext, d_bits, nullptr, st);
        if (rc == CPX_OK) note_kernel("map_exact_kernel<true> (%d states, one codeword per lane)", t->S);
        return rc;
    }
    const int GW = pick_gw(t->S, B);
    const int64_t npairs = (B + GW - 1) / GW;
    const int np = pick_npair(npairs);
    const int64_t nblocks = (npairs + np - 1) / np, K = (N + CH - 1) / CH;
    p.GW = GW;
    p.sys = d_sys; p.par = d_par; p.Lin = d_L_int; p.Lout = d_L_ext; p.bits = d_bits;
    p.B = B; p.N = N; p.nv2 = 2 * noise_variance; p.want_bits = want_bits;
    CPX_REQUIRE(nblocks < (1ll << 31), CPX_ELIMIT, "map_decode: batch too large");
    if ((rc = workspace(st, 0, sizeof(double) * (size_t)(nblocks * np * (K + 1) * 64), (void **)&p.scratch))) return rc;
    // "detect and redo": one flag byte per codeword (scratch-arena slot 3), zeroed here, set by the kernel, consumed by the
    // absolute-scale redo launch below; blocks too long for that path's scratch are decoded by the fast kernel alone
    // (round 5: the redo launch is the wave-parallel literal kernel below, which needs no per-lane scratch: no block-length limit)
    if ((rc = workspace(st, 3, (size_t)B, (void **)&p.flags))) return rc;
    CPX_HIP(hipMemsetAsync(p.flags, 0, (size_t)B, st));
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
    // redo: pairs with a flagged codeword, literally (map_literal_kernel), same launch geometry
    const RedoCounter redo = redo_counter(st);
    switch (p.tb.lgS) {
#define CASE(LG) case LG: hipLaunchKernelGGL((map_literal_kernel<LG, false>), grid, block, sizeof(double) * 2 * np * wave_lds_doubles<LG>(GW), st, p, redo); break;
        case 2:
            if (p.tb.sr4) hipLaunchKernelGGL((map_literal_kernel<2, true>), grid, block, sizeof(double) * 2 * np * wave_lds_doubles<2>(GW), st, p, redo);
            else hipLaunchKernelGGL((map_literal_kernel<2, false>), grid, block, sizeof(double) * 2 * np * wave_lds_doubles<2>(GW), st, p, redo);
            break;
        CASE(1) CASE(3) CASE(4)
#undef CASE
        default: break;
    }
    CPX_HIP(hipGetLastError());
    note_kernel("map_decode_kernel<%d,%s> (%d wave pairs per workgroup, %d codewords per pair) + map_literal_kernel", p.tb.lgS, (p.tb.lgS == 2 && p.tb.sr4) ? "true" : "false", np, GW);
    note_redo((long long)B, "codewords flagged (their pairs decoded literally)");
    return CPX_OK;
}

int cpx_turbo_decode_batch_dev(const cpx_trellis *t, const double *d_sys, const double *d_p1, const double *d_p2,
                               const double *d_L_int_or_null, const int32_t *d_perm, int64_t B, int64_t N,
                               double noise_variance, int n_iter, uint8_t *d_bits, void *stream) {
    CPX_TRACE("cpx_turbo_decode_batch_dev");
    cpx::IssueGuard issue_guard;
    TurboParams p;
    int rc = fill_tables(t, p.tb);
    if (rc) return rc;
    CPX_REQUIRE(B >= 0 && N >= 0 && n_iter >= 0, CPX_EINVAL, "turbo_decode: negative size");
    CPX_REQUIRE(N < (1ll << 21), CPX_ELIMIT, "turbo_decode: blocks of 2^21 steps or more are not supported (31-bit lane offsets into a 16-codeword slab)");
    if (B == 0 || N == 0) return CPX_OK;
    hipStream_t st = pick_stream(stream);
    if (t->S > 16) {                                              // see cpx_map_decode_batch_dev
        rc = bcjr_exact_turbo(t, d_sys, d_p1, d_p2, d_L_int_or_null, d_perm, B, N, 2 * noise_variance, n_iter, d_bits, nullptr, st);
        if (rc == CPX_OK) note_kernel("turbo_exact_kernel<true> (%d states, one codeword per lane)", t->S);
        return rc;
    }
    const int GW = pick_gw(t->S, B);
    const int64_t npairs = (B + GW - 1) / GW;
    const int np = pick_npair(npairs);
    const int64_t nblocks = (npairs + np - 1) / np, K = (N + CH - 1) / CH;
    p.GW = GW;
    p.sys = d_sys; p.p1 = d_p1; p.p2 = d_p2; p.Lint = d_L_int_or_null; p.perm = d_perm; p.bits = d_bits;
    p.B = B; p.N = N; p.nv2 = 2 * noise_variance; p.n_iter = n_iter;
    CPX_REQUIRE(nblocks < (1ll << 31), CPX_ELIMIT, "turbo_decode: batch too large");
    if ((rc = workspace(st, 0, sizeof(double) * (size_t)(nblocks * np * (K + 1) * 64), (void **)&p.ckpt))) return rc;
    // the time-major slab (TurboParams): five arrays of N rows of 16 slots per row group; sized in float64 whatever the precision
    // mode (the literal redo kernel reuses a pair's share as float64 scratch)
    const int RW = (int)TM_RW;
    p.RW = RW;
    p.lgG = 0;
    while ((GW << p.lgG) < RW) p.lgG++;
    const int64_t ngroups = (npairs + (1 << p.lgG) - 1) >> p.lgG;
    p.NR = N | 1;
    CPX_REQUIRE(5 * p.NR * RW * 8 < (1ll << 31), CPX_ELIMIT, "turbo_decode: block too long for 31-bit lane offsets into a row group's slab");
    if ((rc = workspace(st, 1, sizeof(double) * (size_t)(ngroups * 5 * p.NR * RW), (void **)&p.larr))) return rc;
    // "detect and redo", as in cpx_map_decode_batch_dev (round 5: redone by turbo_literal_kernel, no block-length limit)
    if ((rc = workspace(st, 3, (size_t)B, (void **)&p.flags))) return rc;
    CPX_HIP(hipMemsetAsync(p.flags, 0, (size_t)B, st));
    CPX_REQUIRE(npairs < (1ll << 31), CPX_ELIMIT, "turbo_decode: batch too large");
    // the two row index tables the library derives from the interleaver
    {
        int32_t *tabs = nullptr;
        if ((rc = workspace(st, 11, 2 * sizeof(int32_t) * (size_t)N, (void **)&tabs))) return rc;
        p.iperm = tabs; p.ident = tabs + N;
        hipLaunchKernelGGL(turbo_tables_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, d_perm, tabs, tabs + N, N);
    }
    if (n_iter == 0) {                                            // the reference's loop never runs: no decision is ever taken
        CPX_HIP(hipMemsetAsync(d_bits, 0, (size_t)(B * N), st));
        note_kernel("turbo_decode: 0 iterations");
        return CPX_OK;
    }
    dim3 grid((unsigned)nblocks), block(128 * np);
    // "fp32-fast" (cpx_set_precision; not the parity mode): the slab between the passes in float32 (slab_ld), arithmetic unchanged
    const bool s32 = precision_fast();
    const int itiles = (int)((N + IT - 1) / IT), ftiles = (int)((N + FT - 1) / FT);
    CPX_REQUIRE(ngroups * itiles < (1ll << 31) && ngroups * ftiles < (1ll << 31), CPX_ELIMIT, "turbo_decode: batch too large");
    {
        const dim3 igrid((unsigned)(ngroups * itiles));
        const size_t ilds = sizeof(double) * 4 * (size_t)RW * IT_ROW;
        if (s32) hipLaunchKernelGGL(turbo_init_kernel<true>, igrid, dim3(256), ilds, st, p, itiles);
        else hipLaunchKernelGGL(turbo_init_kernel<false>, igrid, dim3(256), ilds, st, p, itiles);
    }
    auto pass = [&](const int32_t *idx, int second, int pout, int llr_in, int last) -> int {
        switch (p.tb.lgS) {
#define LAUNCH(LG, SRV) do { if (s32) hipLaunchKernelGGL((turbo_pass_kernel<LG, SRV, true>), grid, block, sizeof(double) * 2 * np * wave_lds_doubles<LG>(GW), st, p, idx, second, pout, llr_in, last); \
                             else hipLaunchKernelGGL((turbo_pass_kernel<LG, SRV, false>), grid, block, sizeof(double) * 2 * np * wave_lds_doubles<LG>(GW), st, p, idx, second, pout, llr_in, last); } while (0)
#define CASE(LG) case LG: LAUNCH(LG, false); break;
            case 2:
                if (p.tb.sr4) LAUNCH(2, true);
                else LAUNCH(2, false);
                break;
            CASE(1) CASE(3) CASE(4)
#undef CASE
#undef LAUNCH
            default: set_error("turbo_decode: unsupported state count"); return CPX_ELIMIT;
        }
        return CPX_OK;
    };
    static const bool no_pout = [] { const char *e = getenv("CPX_TURBO_POUT"); return e && e[0] == '0'; }();   // A/B runs
    int prev_pout = 1;                                            // (turbo_init_kernel wrote a prior)
    for (int h = 0; h < 2 * n_iter; h++) {
        // every pass but the last two hands prior0(E) to the next one directly (epilogue): the last MAP 2 needs L_int_2 and E_2 as
        // LLRs for the decision L_2 = L_int_2 + E_2 > 0 (:148-152, :326-331), so the last MAP 1 writes E_1 itself
        const int pout = (h <= 2 * n_iter - 3 && !no_pout) ? 1 : 0;
        const int last = h == 2 * n_iter - 1;
        // the prior of MAP 2 is interlv(E_1) (:319): row perm[t]; of MAP 1 deinterlv(E_2) (:329): row inverse_perm[t]; of the very
        // first pass L_int_1 as turbo_init_kernel laid it out: row t
        const int32_t *idx = (h & 1) ? d_perm : (h == 0 ? p.ident : p.iperm);
        if ((rc = pass(idx, h & 1, pout, (h > 0 && !prev_pout) ? 1 : 0, last))) return rc;
        prev_pout = pout;
    }
    if (s32) hipLaunchKernelGGL(turbo_final_kernel<true>, dim3((unsigned)(ngroups * ftiles)), dim3(FT), 0, st, p, ftiles);
    else hipLaunchKernelGGL(turbo_final_kernel<false>, dim3((unsigned)(ngroups * ftiles)), dim3(FT), 0, st, p, ftiles);
    CPX_HIP(hipGetLastError());
    note_kernel("turbo_pass_kernel<%d,%s%s> x %d (time-major slab, interleaver = row index) + turbo_init_kernel + turbo_final_kernel (%d wave pairs per workgroup, %d codewords per pair)", p.tb.lgS,
                (p.tb.lgS == 2 && p.tb.sr4) ? "true" : "false", s32 ? ",f32 slab" : "", 2 * n_iter, np, GW);
    // redo: one launch; a pair with a flagged codeword decodes its codewords again, literally, all iterations (turbo_literal_kernel)
    const RedoCounter redo = redo_counter(st);
    {
        const dim3 lgrid((unsigned)npairs), lblock(128);
        switch (p.tb.lgS) {
#define CASE(LG) case LG: hipLaunchKernelGGL((turbo_literal_kernel<LG, false>), lgrid, lblock, sizeof(double) * 2 * wave_lds_doubles<LG>(GW), st, p, redo); break;
            case 2:
                if (p.tb.sr4) hipLaunchKernelGGL((turbo_literal_kernel<2, true>), lgrid, lblock, sizeof(double) * 2 * wave_lds_doubles<2>(GW), st, p, redo);
                else hipLaunchKernelGGL((turbo_literal_kernel<2, false>), lgrid, lblock, sizeof(double) * 2 * wave_lds_doubles<2>(GW), st, p, redo);
                break;
            CASE(1) CASE(3) CASE(4)
#undef CASE
            default: break;
        }
        CPX_HIP(hipGetLastError());
    }
    note_redo((long long)B, "codewords flagged (their pairs decoded literally)");
    return CPX_OK;
}

int cpx_map_decode_batch(const cpx_trellis *t, const double *sys, const double *par, const double *L_int, int64_t B,
                         int64_t N, double noise_variance, int want_bits, double *L_ext, uint8_t *bits) {
    CPX_TRACE("cpx_map_decode_batch");
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
    if ((rc = d2h_pageable(L_ext, dout.p, nb, st))) return rc;
    CPX_HIP(hipMemcpyAsync(bits, dbits.p, (size_t)(B * N), hipMemcpyDeviceToHost, st));
    CPX_HIP(hipStreamSynchronize(st));
    return CPX_OK;
}

int cpx_turbo_decode_batch(const cpx_trellis *t, const double *sys, const double *p1, const double *p2,
                           const double *L_int_or_null, const int32_t *perm, int64_t B, int64_t N,
                           double noise_variance, int n_iter, uint8_t *bits) {
    CPX_TRACE("cpx_turbo_decode_batch");
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
