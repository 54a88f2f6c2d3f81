// LDPC belief propagation with the whole decoder state of a block resident in LDS (gfx950: 160 KB per CU).
// Same contract and the same float64 operations per edge, in the same order, as ldpc.hip -- which documents the formulation
//   ldpc_bp_decode  (/root/reference/commpy/channelcoding/ldpc.py:144-254)
// and stays the path for codes whose state does not fit (and for n_iters == 0).
//
// Why.  The tiled path keeps R/Q/L of 64 blocks per wavefront in HBM and runs one launch per pass: an iteration gathers
// 2 E rows per tile through L2 (latency-bound: 0.4 + 0.6 ms per iteration at B = 32 768), frozen blocks hold lanes until a
// move pass compacts the working set, and a decode is 4 launches x n_iters.  Here ONE persistent launch does everything:
//   * a workgroup decodes ONE block at a time: the a-posteriori LLRs Q[n_v] and one check->variable message per edge,
//     R[check][position] with a fixed odd row stride (the largest check degree: 11 for (1944,1296)), live in LDS for the
//     block's whole life (72.6 KB, two workgroups per compute unit).  An even stride, e.g. the 12 doubles of the padded
//     tables, puts every 8th lane of a wavefront on the same LDS banks: measured 3.1 instead of 2.5 ms;
//   * thread = node.  The check pass walks the checks, the variable pass the variables; both gather from LDS (~100 cycles)
//     instead of L2, through tables of pre-scaled LDS byte offsets (no address arithmetic in the loops).  Only the channel
//     LLR of the variable pass (`+ llr`, :245) is re-read from HBM/L2: 8 B per variable and iteration;
//   * rows and columns are padded to a multiple of four entries that point at two dummy slots -- Q = +inf, R = +0.0 --
//     which are neutral for every accumulation of the passes (min, sign parity, syndrome parity, column sum), and a padded
//     row entry has its own (never gathered) R slot: the min-sum loops carry no per-edge predicate at all;
//   * when its block's syndrome is zero (:203-206), or after n_iters iterations, the workgroup retires the block to a
//     block-major staging buffer and takes the next one from a global queue: continuous batching replaces the scan / move /
//     compaction kernels, and nobody waits for the slowest block of a tile.  A zero-initialised R makes the first
//     iteration exact without a special case: the first variable->check message is R * -1 + 1.0 * Q = -0.0 + Q = Q
//     bit for bit (:199 vs :244-245);
//   * min-sum stores the messages themselves (the tiled path stores a 3-word record per check and regenerates them: same
//     values, fewer bytes, more instructions -- with the records the first resident version ran at 88 % VALU busy and
//     6 080 instructions per block-iteration; this one needs 2 900 and is bound by LDS bandwidth, 82 % busy);
//   * a final transpose kernel turns the staging buffer [B][n_v] into the reference layout [n_v][B] (:251-253) and
//     writes dec_word.
// HBM traffic per block: llr read once per executed iteration (L2 hits after the first), staging written once.
#include "cpx_internal.h"
#include "cpx_math.h"
#include "ldpc_dev.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <mutex>

using namespace cpx;

namespace {

constexpr size_t LDS_BYTES = 160 * 1024;

struct ResParams {
    double *llr;             // [B][n_v], clipped in place (:186)
    double *out;             // [B][n_v] a-posteriori LLRs of retired blocks, one block per ROW: the caller's block-major out_llrs, or
                             // the staging buffer that ldpc_unstage_kernel transposes into the [n_v][B] layout
    int8_t *dec;             // [B][n_v] dec_word of retired blocks (block-major outputs), or null (the transpose kernel writes it)
    int spa_exact;           // sum-product: every row by the exact-order sequence (CPX_LDPC_SPA=exact; A/B and strict runs)
    int32_t *iters;          // [B] executed iterations, may be null
    int *queue;              // next block to hand out
    int *clipped;            // set to 1 when the in-place clip changed a value (may be null)
    uint8_t *nanflags;       // min-sum: [B], 1 = a NaN among the block's LLRs (decoded again by ldpc_exact_kernel<false>); else null
    const int32_t *row_deg;  // [n_c] check degree
    const int32_t *row_q;    // [n_c][cpad] LDS byte offset of Q[variable of the j-th edge]; padding -> the +inf slot
    const int32_t *col_r;    // [n_v][vpad] LDS byte offset of R[q-th edge of the variable], increasing check; padding -> the 0.0 slot
    const int32_t *vgrp;     // [ceil(n_v / 64)] chunks of four column entries the 64 variables of a wavefront need
    int64_t B;
    int n_r;                 // R slots = n_c * rstride (+ 1 dummy + slack for the padded reads of the last row)
    int rstride;             // doubles per row of R: max check degree, made odd (an even stride of 8-byte words is an LDS bank conflict)
    int n_v, n_c, cpad, vpad, max_iter;
    int roff, ctl_off;       // LDS byte offsets of R and of the control words
    double *e0;              // ldpc_resident_ratio_kernel: [workgroups][n_v] exp(llr) of the block a workgroup is decoding (L2-resident scratch)
};

// LDS accesses by absolute byte address: the kernel declares no static LDS, so its dynamic segment starts at address 0
// (checked at kernel entry) and the offset tables hold ready-made addresses -- no base add in front of every ds_read.
typedef __attribute__((address_space(3))) double lds_f64;
__device__ __forceinline__ double ldsd(int off) { return *reinterpret_cast<lds_f64 *>((unsigned)off); }
__device__ __forceinline__ void stsd(int off, double v) { *reinterpret_cast<lds_f64 *>((unsigned)off) = v; }

// ---- min-sum check node (:229-238 after :244-245): m_j = Q[v_j] - R_j;  R_j <- prod_{i != j} sign(m_i) * min_{i != j} |m_i| ----
// CQ > 0: rows of at most 4 CQ entries, fully unrolled.  The messages are written from (min1, min2, argmin, signs), exactly
// sign(other).prod() * abs(other).min(): a zero among the others makes the minimum zero by itself.
// `held`: the row's offset table already in registers (CQ > 0, see the kernel), else null
template <int CQ>
__device__ __forceinline__ void check_msa(const ResParams &p, int c, int *flag, const int4 *held = nullptr) {
    const int4 *__restrict__ qv = reinterpret_cast<const int4 *>(p.row_q + (int64_t)c * p.cpad);
    const int rb = p.roff + 8 * c * p.rstride;
    int sx = 0, imin = 0;
    unsigned neg = 0;                                            // sign bits, shifted in from the right: edge j ends at bit N - 1 - j
    double m1 = __builtin_huge_val(), m2 = __builtin_huge_val();
    constexpr int NQ = CQ > 0 ? CQ : 1;
    // One edge: seven instructions.  |m| never exists as a value -- the compare and the min / max take it as an operand modifier --;
    // the second minimum needs no select: the candidate is the LARGER of (first minimum so far, |m|) either way; the sign goes into
    // `neg` with one v_alignbit (the sign BIT: a message of -0.0 counts as negative, which only changes the sign of zero messages
    // -- never observable: a column sum starts from +0.0 and a row with a zero sends +-0.0 to everyone else).  Round 2 spent 14:
    // |m| materialised (and + mov), a 64-bit select for the second minimum, compare / select / or for the sign, with s_nops between.
#define CPX_MSA_IN(j, q, JC)                                                                              \
    {                                                                                                  \
        const double m = ldsd(rb + 8 * (j)) * -1.0 + (q);        /* data * -1 + 1.0 * (msg_sum + llr) (:244-245); first pass -0.0 + q (:199) */ \
        double big;                                                                                    \
        /* one statement: three instructions sit between the float64 compare and the select that reads its mask (back to back */ \
        /* the pair costs two wait states), and the old first minimum is read by compare and max before the min overwrites it */ \
        asm("v_cmp_lt_f64 vcc, |%5|, %0\n\tv_max_f64 %4, %0, |%5|\n\tv_min_f64 %0, %0, |%5|\n\t"               \
            "v_alignbit_b32 %2, %2, %6, 31\n\tv_cndmask_b32_e64 %3, %3, %7, vcc\n\tv_min_f64 %1, %1, %4"      \
            : "+v"(m1), "+v"(m2), "+v"(neg), "+v"(imin), "=&v"(big)                                     \
              : "v"(m), "v"(__double2hiint(m)), JC(j) : "vcc");                                        \
    }
    // four edges: the gathers, the syndrome parity of their sign words (dec_word = out_llrs < 0, :193, :248) as two 3-input xors
#define CPX_MSA_IN4(t, a, JC)                                                                            \
    {                                                                                                  \
        const double q0 = ldsd((a).x), q1 = ldsd((a).y), q2 = ldsd((a).z), q3 = ldsd((a).w);           \
        asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(sx) : "v"(__double2hiint(q0)), "v"(__double2hiint(q1))); \
        asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(sx) : "v"(__double2hiint(q2)), "v"(__double2hiint(q3))); \
        CPX_MSA_IN(4 * (t) + 0, q0, JC) CPX_MSA_IN(4 * (t) + 1, q1, JC) CPX_MSA_IN(4 * (t) + 2, q2, JC) CPX_MSA_IN(4 * (t) + 3, q3, JC) \
    }
    int N;                                                       // entries shifted into `neg`
    if (CQ > 0) {
        int4 a[NQ];
#pragma unroll
        for (int t = 0; t < NQ; t++) a[t] = held ? held[t] : qv[t];
#pragma unroll
        for (int t = 0; t < NQ; t++) CPX_MSA_IN4(t, a[t], "n")    // edge number: an inline constant
        N = 4 * NQ;
    } else {
        const int nq = p.cpad >> 2;
        for (int t = 0; t < nq; t++) {
            const int4 a = qv[t];
            CPX_MSA_IN4(t, a, "v")                               // ... a register
        }
        N = 4 * nq;
    }
#undef CPX_MSA_IN4
#undef CPX_MSA_IN
    if (sx < 0) *flag = 1;                                       // odd row: this iteration is executed (:205)
    // every entry of the row (padding included: its slot is never gathered) gets +-min1, then the argmin's own entry +-min2
    const unsigned negp = (__popc(neg) & 1) ? ~neg : neg;        // bit N-1-j: sign of the product of the OTHER messages
    const int h1 = __double2hiint(m1), l1 = __double2loint(m1);
#define CPX_MSA_OUT(j) stsd(rb + 8 * (j), __hiloint2double(h1 | (int)((negp << (32 - N + (j))) & 0x80000000u), l1));
    if (CQ > 0) {
#pragma unroll
        for (int j = 0; j < 4 * NQ - 4; j++) CPX_MSA_OUT(j)      // rstride > 4 (NQ - 1): these positions always belong to the row
#pragma unroll
        for (int j = 4 * NQ - 4; j < 4 * NQ; j++)
            if (j < p.rstride) CPX_MSA_OUT(j)                    // wave-uniform: positions past the stride are the next row's
    } else {
        for (int j = 0; j < p.rstride; j++) CPX_MSA_OUT(j)
    }
#undef CPX_MSA_OUT
    stsd(rb + 8 * imin, __hiloint2double(__double2hiint(m2) | (int)((negp << (32 - N + imin)) & 0x80000000u), __double2loint(m2)));
}

// ---- sum-product check node (:209-227): one division per edge, exact-order redo of rows near saturation (ldpc_dev.h);
// the row keeps e_j = exp(-|m_j|) with the sign of m_j in its own R entries between the two loops --------------------------
__device__ __forceinline__ void check_spa(const ResParams &p, int c, int *flag) {
    const int deg = p.row_deg[c];
    const int4 *__restrict__ qv = reinterpret_cast<const int4 *>(p.row_q + (int64_t)c * p.cpad);
    const int rb = p.roff + 8 * c * p.rstride;
    int sx = 0;
    double U = 1.0, W = 1.0, emax = 0.0;
    for (int j0 = 0; __builtin_amdgcn_ballot_w64(j0 < deg) != 0; j0 += 4) {
        const int4 a = qv[j0 >> 2];
        const double q[4] = {ldsd(a.x), ldsd(a.y), ldsd(a.z), ldsd(a.w)};
#pragma unroll
        for (int u4 = 0; u4 < 4; u4++) {
            const int j = j0 + u4;
            if (j < deg) {
                sx ^= __double2hiint(q[u4]);                     // dec_word = out_llrs < 0 (:193, :248)
                double m = ldsd(rb + 8 * j) * -1.0;              // data *= -1 (:244); first pass: 0 * -1 = -0.0
                m += 1.0 * q[u4];                                // data += H.multiply(msg_sum + llr).data (:245); first pass (:199)
                double se, u, w;
                spa_in(m, se, u, w);                             // e = exp(-|m|): tanh(m / 2) = u / w (:210-211)
                U *= u; W *= w;                                  // row product (reference: exp2(sum(log2)) :217-219)
                emax = fmax(emax, fabs(se));                     // the edge with the smallest |m| (NaN: spa_row_near sees U)
                stsd(rb + 8 * j, se);
            }
        }
    }
    if (sx < 0) *flag = 1;
    const bool near = p.spa_exact || spa_row_near(U, W, emax);   // see ldpc_dev.h: near rows take the exact-order sequence
    if (__builtin_amdgcn_ballot_w64(!near) != 0) {
        for (int j = 0; __builtin_amdgcn_ballot_w64(j < deg) != 0; j++)
            if (j < deg && !near) stsd(rb + 8 * j, spa_out_fast(U, W, ldsd(rb + 8 * j)));
    }
    if (__builtin_amdgcn_ballot_w64(near) != 0) {
        double prod = 1.0;
        for (int j = 0; __builtin_amdgcn_ballot_w64(j < deg) != 0; j++)
            if (j < deg && near) prod *= spa_exact_t(ldsd(rb + 8 * j));
        for (int j = 0; __builtin_amdgcn_ballot_w64(j < deg) != 0; j++)
            if (j < deg && near) stsd(rb + 8 * j, spa_out_exact(spa_exact_t(ldsd(rb + 8 * j)), prod));
    }
}

// ---- variable node, both algorithms: column sum in increasing check order + llr (:243-247) ----------
// `a0`, `l`: the first four column entries and the channel LLR of the variable, requested by the caller one round ahead (the
// first round's before the barrier that ends the check pass): a variable of degree <= 4 -- most of them -- starts its LDS gathers
// without waiting for memory.
__device__ __forceinline__ void var_node(const ResParams &p, int v, int4 a0, double l) {
    const int trips = p.vgrp[__builtin_amdgcn_readfirstlane(v) >> 6];      // the lanes of a wavefront hold 64 consecutive variables
    const int4 *__restrict__ rf = reinterpret_cast<const int4 *>(p.col_r + (int64_t)v * p.vpad);
    double msum = 0.0;
    {
        const double r0 = ldsd(a0.x), r1 = ldsd(a0.y), r2 = ldsd(a0.z), r3 = ldsd(a0.w);
        msum += r0; msum += r1; msum += r2; msum += r3;          // message_matrix.sum(0); padding adds +0.0 to a sum that is never -0.0
    }
    for (int t = 1; t < trips; t++) {
        const int4 a = rf[t];
        const double r0 = ldsd(a.x), r1 = ldsd(a.y), r2 = ldsd(a.z), r3 = ldsd(a.w);
        msum += r0; msum += r1; msum += r2; msum += r3;
        // (skipping the entries nobody in the wavefront has -- wave-uniform branches around the reads, same for the 12th row
        //  position in check_msa -- saves 10 % of the LDS traffic and measured 3.55 instead of 2.30 ms: the branches fence the
        //  scheduler and every read waits alone)
    }
    stsd(8 * v, msum + l);                                       // msg_sum + llr (:245, :247)
}

template <int ALG, int CQ>
__global__ __launch_bounds__(1024, 6) void ldpc_resident_kernel(ResParams p) {
    extern __shared__ __align__(16) char lds[];
    if ((unsigned)(uintptr_t)lds != 0u) __builtin_trap();        // the tables hold absolute LDS addresses (see ldsd)
    int *ctl = reinterpret_cast<int *>(lds + p.ctl_off);         // [0], [1]: "unsatisfied" flag of even / odd iterations; [2]: block
    const int tid = threadIdx.x, nt = blockDim.x;
    // Control flow around the barriers is kept UNIFORM on purpose: the block index and the syndrome flag are read through
    // readfirstlane (scalar loop exits), and everything only thread 0 does -- retiring a block, taking the next one from the
    // queue -- sits in ONE if-block in the middle of the loop body, followed by a barrier.  With a thread-0 block at the end of
    // the body and another at its start the structuriser turned the loop inside out (lanes 1..63 of wave 0 went round to the
    // next s_barrier while lane 0 was still retiring): wave 0 met the barrier twice per block and the kernel hung.
    // (Round 5 measured popping the queue one block AHEAD -- the atomic's round trip off the critical path -- and lost: the extra live
    // register spills in the sum-product kernels (+5 %), min-sum +1 %; the second workgroup on the compute unit already hides the
    // round trip.  profiles/r05_ldpc_pop_hoist_ab.txt)
    auto pop = [&]() {                                           // thread 0 only
        const int t = atomicAdd(p.queue, 1);
        ctl[2] = t < p.B ? t : -1;
    };
    if (tid == 0) {
        stsd(8 * p.n_v, __builtin_huge_val());                   // dummy Q (row padding)
        // dummy R (column padding) and the slack behind it: the padded entries of the LAST row read up to three slots past its
        // stride.  They must hold numbers: a padded entry is m = -R + inf, and min-sum's check node (v_max / v_min return the
        // other operand for a NaN) is only right when that is +inf, never NaN (uninitialised LDS: a wrong second minimum and
        // a flipped row parity whenever the previous kernel had left a NaN pattern there)
        for (int i = 0; i < 5; i++) stsd(p.roff + 8 * (p.n_r + i), 0.0);
        ctl[0] = 0; ctl[1] = 0;                                  // "unsatisfied" flags of the first block
        ctl[3] = 0;                                              // "a NaN among the LLRs of the current block"
        pop();
    }
    __syncthreads();
    // Min-sum with unrolled rows, all checks in one round (thread = check for the whole launch): the row's table of Q addresses
    // stays in registers.  Read from memory in every check pass it was three dependent 16-byte loads in front of the first LDS
    // gather of every wave and iteration, and 31 KB per block-iteration through the vector L1.
    constexpr int HQ = (ALG == CPX_LDPC_MSA && CQ > 0 && CQ <= 3) ? CQ : 1;   // (rows of 13 .. 16 entries: the 16 registers would spill)
    const bool held_rows = ALG == CPX_LDPC_MSA && CQ > 0 && CQ <= 3 && p.n_c <= nt;
    int4 hq[HQ];
#pragma unroll
    for (int t = 0; t < HQ; t++) hq[t] = make_int4(0, 0, 0, 0);
    if (held_rows && tid < p.n_c) {
#pragma unroll
        for (int t = 0; t < HQ; t++) hq[t] = reinterpret_cast<const int4 *>(p.row_q + (int64_t)tid * p.cpad)[t];
    }
    int b = __builtin_amdgcn_readfirstlane(ctl[2]);
    while (b >= 0) {
        double *__restrict__ in = p.llr + (int64_t)b * p.n_v;
        // four LLRs per thread are requested before the first is used (round 5: one at a time, a block began with ceil(n_v / nt)
        // memory latencies in a row -- three for (1944,1296) -- with the whole workgroup waiting behind the barrier below)
        for (int v0 = tid; v0 < p.n_v; v0 += 4 * nt) {
            double rawv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) rawv[u] = in[v0 + u * nt < p.n_v ? v0 + u * nt : v0];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int v = v0 + u * nt;
                if (v >= p.n_v) break;
                const double raw = rawv[u];
                const double x = clip_nan(raw, -500.0, 500.0);
                if (x != raw) {                                  // in-place clip (:186); untouched values are not rewritten
                    if (x == x) {
                        in[v] = x;
                        if (p.clipped) *p.clipped = 1;
                    } else {
                        ctl[3] = 1;                              // NaN: min-sum decodes the block again (see ResParams::nanflags)
                    }
                }
                stsd(8 * v, x);                                  // out_llrs = llr (:194)
            }
        }
        for (int e = tid; e < p.n_r; e += nt) stsd(p.roff + 8 * e, 0.0);
        __syncthreads();
        int k = 0;
        for (; k < p.max_iter; k++) {
            int *flag = &ctl[k & 1];
            if (ALG == CPX_LDPC_MSA && held_rows) {
                if (tid < p.n_c) check_msa<CQ>(p, tid, flag, hq);
            } else {
                for (int c = tid; c < p.n_c; c += nt) {
                    if (ALG == CPX_LDPC_MSA) check_msa<CQ>(p, c, flag);
                    else check_spa(p, c, flag);
                }
            }
            // min-sum: the variable pass's first table entries and channel LLR are requested before the barrier (see var_node);
            // every variable has a padded table row, so the address is valid for tid < n_v -- else entry 0 is read and not used.
            // (Sum-product sits at its 80-register cap: the six extra registers spilled and cost 1 %; it loads in the loop.)
            constexpr bool AHEAD = ALG == CPX_LDPC_MSA;
            int4 nx = make_int4(0, 0, 0, 0);
            double nl = 0.0;
            if (AHEAD) {
                nx = reinterpret_cast<const int4 *>(p.col_r + (int64_t)(tid < p.n_v ? tid : 0) * p.vpad)[0];
                nl = in[tid < p.n_v ? tid : 0];
            }
            __syncthreads();
            if (!__builtin_amdgcn_readfirstlane(*flag)) break;   // zero syndrome: the block keeps the Q it has (:205-206)
            if (tid == 0) ctl[(k + 1) & 1] = 0;
            for (int v = tid; v < p.n_v; v += nt) {
                if (AHEAD) {
                    const int4 a0 = nx;
                    const double l = nl;
                    const int vn = v + nt < p.n_v ? v + nt : v;  // (the last round requests its own entry again: no branch around loads)
                    nx = reinterpret_cast<const int4 *>(p.col_r + (int64_t)vn * p.vpad)[0];
                    nl = in[vn];
                    var_node(p, v, a0, l);
                } else {
                    var_node(p, v, reinterpret_cast<const int4 *>(p.col_r + (int64_t)v * p.vpad)[0], in[v]);
                }
            }
            __syncthreads();
        }
        // Retire: the a-posteriori LLRs (and, for block-major outputs, their sign bits) as one contiguous row -- the layout the
        // reference's own results have in memory (ldpc.py:251-253 reshapes with order='F': one block per column of an F-ordered
        // array IS one block per contiguous row).  Not waited for: the next block's loads queue behind these stores.
        double *__restrict__ orow = p.out + (int64_t)b * p.n_v;
        if (p.dec) {
            int8_t *__restrict__ drow = p.dec + (int64_t)b * p.n_v;
            for (int v = tid; v < p.n_v; v += nt) {
                const double x = ldsd(8 * v);                            // the same thread reloads these entries for the next block
                orow[v] = x;                                     // (:247)
                drow[v] = (int8_t)(__builtin_signbit(x) ? 1 : 0);    // (:248)
            }
        } else {
            for (int v = tid; v < p.n_v; v += nt) orow[v] = ldsd(8 * v);
        }
        if (tid == 0) {
            if (p.iters) p.iters[b] = k;
            if (p.nanflags) p.nanflags[b] = (uint8_t)ctl[3];
            ctl[3] = 0;
            ctl[0] = 0; ctl[1] = 0;                              // (a thread that has yet to read a flag of this block reads the 0 that sent the others here)
            pop();
        }
        __syncthreads();
        b = __builtin_amdgcn_readfirstlane(ctl[2]);
    }
}

// ---- sum-product in the RATIO domain (round 4) ---------------------------------------------------------------------------------------
// The log-domain row above spends most of its instructions on one exp (e = exp(-|m|)) and one log (2 atanh) per edge and iteration.
// Kept as likelihood ratios -- X_v = exp(out_llr_v) per variable, rho_cj = exp(R_cj) per edge -- the same iteration needs neither:
//     exp(m_j) = X[v_j] / rho_j,    e_j = exp(-|m_j|) = min(X, rho) / max(X, rho)  (ONE division),    sign(m_j) = (X >= rho)
//     rho_j  <-  (W u_j + U w_j) / (W u_j - U w_j)        (the one-division row of ldpc_dev.h without its logarithm)
//     X_v    <-  exp(llr_v) prod_j rho_j                   (exp(llr) once per block, kept in an L2-resident scratch row)
// and out_llrs = log X is formed ONCE, when the block retires.  The sign of out_llr_v (dec_word, the syndrome test) is X_v < 1: it is
// kept in the sign bit of the stored X.  The arithmetic differs from the log-domain row by roundings of ~1e-16 relative to the
// ratios, i.e. ~1e-15 absolute on the messages: the same class as the one-division row (scripts/micro/spa_ratio_emul.py decodes the 72
// live-reference blocks of ldpc_c4y.npz with it in NumPy: dec_word, iteration counts, banded contract all met; near rows 0.03 - 0.7 %).
// Range: the product of a variable is taken over its factors above and below 1 separately; if either leaves the normal range, or the
// ratio leaves (2^-990, 2^990), the variable is summed through logarithms and, beyond +-680, stored as the LLR itself (ratio_encode).
// What the ratio domain cannot carry is decoded again at once, by the same workgroup, from the untouched LLRs, with the log-domain row
// (spa_block_log: the loop of ldpc_resident_kernel<SPA>; a second launch for these blocks ended on its longest block, 0.2 - 0.6 ms of a
// 4.6 - 8 ms decode, here the queue absorbs them like any other slow block):
//   * a NaN (an LLR of exactly zero makes one with a SIGN the reference's dec_word reads, see ldpc_dev.h; a NaN input);
//   * an iteration in which more than half of the rows are near saturation: there the result is decided by how the reference's own
//     operation sequence rounds (+-500 against 37.4), which the exact-order row reproduces only from log-domain inputs.  Correctly
//     scaled channel LLRs do this to about one block in a hundred (their last iteration); LLRs scaled up several times do it always.
__device__ __forceinline__ double spa_ratio_fast(double U, double W, double se) {
    const double e = fabs(se);
    const double n1 = W * __builtin_copysign(1.0 - e, se), n2 = U * (1.0 + e);
    return div_nr(n1 + n2, n1 - n2);                             // exp(2 atanh(x)) = (1 + x) / (1 - x), |x| < 1 - 2^-32: positive, normal
}

__device__ __forceinline__ void check_spa_ratio(const ResParams &p, int c, int *ctl, int *flag, int *near_rows) {
    const int deg = p.row_deg[c];
    const int4 *__restrict__ qv = reinterpret_cast<const int4 *>(p.row_q + (int64_t)c * p.cpad);
    const int rb = p.roff + 8 * c * p.rstride;
    int sx = 0;
    double U = 1.0, W = 1.0, emax = 0.0;
    for (int j0 = 0; __builtin_amdgcn_ballot_w64(j0 < deg) != 0; j0 += 4) {
        const int4 a = qv[j0 >> 2];
        const double q[4] = {ldsd(a.x), ldsd(a.y), ldsd(a.z), ldsd(a.w)};
#pragma unroll
        for (int u4 = 0; u4 < 4; u4++) {
            const int j = j0 + u4;
            if (j < deg) {
                sx ^= __double2hiint(q[u4]);                     // sign bit of the stored X = (out_llr < 0) (:193, :248)
                const double X = fabs(q[u4]), r = ldsd(rb + 8 * j);
                const double e = div_nr0(min_f64(X, r), max_f64(X, r));     // exp(-|m|), m = out_llr - R (:244-245); 2e-15 on m
                const double se = X >= r ? e : -e;               // ... with the sign of m
                U *= __builtin_copysign(1.0 - e, se);
                W *= 1.0 + e;
                emax = max_f64(emax, e);
                stsd(rb + 8 * j, se);
            }
        }
    }
    if (sx < 0) *flag = 1;
    const bool near = spa_row_near(U, W, emax);
    if (__builtin_amdgcn_ballot_w64(!near) != 0) {
        for (int j = 0; __builtin_amdgcn_ballot_w64(j < deg) != 0; j++)
            if (j < deg && !near) stsd(rb + 8 * j, spa_ratio_fast(U, W, ldsd(rb + 8 * j)));
    }
    if (__builtin_amdgcn_ballot_w64(near) != 0) {
        if (near) atomicAdd(near_rows, 1);
        double prod = 1.0;
        for (int j = 0; __builtin_amdgcn_ballot_w64(j < deg) != 0; j++)
            if (j < deg && near) prod *= spa_exact_t(ldsd(rb + 8 * j));
        for (int j = 0; __builtin_amdgcn_ballot_w64(j < deg) != 0; j++)
            if (j < deg && near) {
                const double R = spa_out_exact(spa_exact_t(ldsd(rb + 8 * j)), prod);
                if (R != R) ctl[3] = 1;                          // the block goes back to the log-domain kernel
                stsd(rb + 8 * j, exp(R));                        // |R| <= 500
            }
    }
}

// X beyond 2^+-990 is past every use as a ratio (tanh(m / 2) is 1.0 in float64 from |m| ~ 37 on): such a slot carries out_llr ITSELF,
// Q 2^1000 above and 2^-1000 / |Q| below, so that a retiring block still has the reference's msg_sum + llr (:245-247) for it
constexpr double RATIO_TOP = 0x1p990, RATIO_BOT = 0x1p-990, RATIO_ENC = 0x1p1000, RATIO_DEC = 0x1p-1000;

__device__ __forceinline__ double ratio_encode(double Q) {       // |Q| >= 680
    return Q > 0.0 ? Q * RATIO_ENC : RATIO_DEC / -Q;
}
__device__ __forceinline__ double ratio_llr(double a) {          // a = |stored X|
    if (a >= RATIO_TOP) return a * RATIO_DEC;
    if (a <= RATIO_BOT) return -(RATIO_DEC / a);
    return fast_log(a);
}

__device__ __forceinline__ void var_node_ratio(const ResParams &p, int v, double e0, const double *__restrict__ in) {
    const int trips = p.vgrp[__builtin_amdgcn_readfirstlane(v) >> 6];
    const int4 *__restrict__ rf = reinterpret_cast<const int4 *>(p.col_r + (int64_t)v * p.vpad);
    double big = max_f64(e0, 1.0), small = min_f64(e0, 1.0);
    for (int t = 0; t < trips; t++) {
        const int4 a = rf[t];
        const double r0 = ldsd(a.x), r1 = ldsd(a.y), r2 = ldsd(a.z), r3 = ldsd(a.w);      // padding: 1.0
        big *= max_f64(r0, 1.0); small *= min_f64(r0, 1.0);
        big *= max_f64(r1, 1.0); small *= min_f64(r1, 1.0);
        big *= max_f64(r2, 1.0); small *= min_f64(r2, 1.0);
        big *= max_f64(r3, 1.0); small *= min_f64(r3, 1.0);
    }
    double X = big * small;
    if (!(big <= 1.7e308) || !(small >= 1e-290) || !(X < RATIO_TOP) || !(X > RATIO_BOT)) {
        // a factor product left the normal range, or the ratio its own: the column sum in the log domain (rare)
        double msum = 0.0;
        for (int t = 0; t < trips; t++) {
            const int4 a = rf[t];
            msum += fast_log(ldsd(a.x)); msum += fast_log(ldsd(a.y)); msum += fast_log(ldsd(a.z)); msum += fast_log(ldsd(a.w));
        }
        const double Q = msum + in[v];
        X = fabs(Q) < 680.0 ? exp(Q) : ratio_encode(Q);          // (a NaN takes the encode branch and stays a NaN: the block is handed back)
    }
    stsd(8 * v, X < 1.0 ? -X : X);
}

// One block in the log domain inside the ratio kernel: the block loop of ldpc_resident_kernel<SPA, 0> (same functions, same order).
// Returns the executed iterations; the a-posteriori LLRs are left in the Q slots.
__device__ __forceinline__ int spa_block_log(const ResParams &p, const double *__restrict__ in, int *ctl, int tid, int nt) {
    for (int v = tid; v < p.n_v; v += nt) stsd(8 * v, clip_nan(in[v], -500.0, 500.0));       // (clipped in place by the first attempt; a NaN stays)
    for (int e = tid; e < p.n_r; e += nt) stsd(p.roff + 8 * e, 0.0);
    if (tid == 0) {
        for (int i = 0; i < 5; i++) stsd(p.roff + 8 * (p.n_r + i), 0.0);                      // column padding: neutral in a SUM
        ctl[0] = 0; ctl[1] = 0;
    }
    __syncthreads();
    int k = 0;
    for (; k < p.max_iter; k++) {
        int *flag = &ctl[k & 1];
        for (int c = tid; c < p.n_c; c += nt) check_spa(p, c, flag);
        __syncthreads();
        if (!__builtin_amdgcn_readfirstlane(*flag)) break;
        if (tid == 0) ctl[(k + 1) & 1] = 0;
        for (int v = tid; v < p.n_v; v += nt) var_node(p, v, reinterpret_cast<const int4 *>(p.col_r + (int64_t)v * p.vpad)[0], in[v]);
        __syncthreads();
    }
    return k;
}

__global__ __launch_bounds__(1024, 6) void ldpc_resident_ratio_kernel(ResParams p) {
    extern __shared__ __align__(16) char lds[];
    if ((unsigned)(uintptr_t)lds != 0u) __builtin_trap();        // absolute LDS addresses (see ldsd)
    // control words: [0], [1] "unsatisfied" flag of even / odd iterations; [2] block; [3] hand the block back (NaN); [5], [6] rows near
    // saturation in the check pass of an even / odd iteration -- every word is written on one side of a barrier and read on the other
    // (see the note on uniform control flow in ldpc_resident_kernel): the counter of iteration k is read by everybody right after the
    // barrier that ends check pass k and zeroed by thread 0 one barrier later, while check pass k + 1 counts into the other one.
    // (Round 4 published "more than half" one iteration late: a block whose LAST executed iteration saturated retired from ratio-domain
    // state, and every handed-back block ran one wasted check pass.)
    int *ctl = reinterpret_cast<int *>(lds + p.ctl_off);
    const int tid = threadIdx.x, nt = blockDim.x;
    double *__restrict__ e0row = p.e0 + (int64_t)blockIdx.x * p.n_v;
    auto pop = [&]() {                                           // thread 0 only
        const int t = atomicAdd(p.queue, 1);
        ctl[2] = t < p.B ? t : -1;
    };
    if (tid == 0) {
        stsd(8 * p.n_v, 1.0);                                    // dummy X (row padding; never read: the rows loop to their degree)
        for (int i = 0; i < 5; i++) stsd(p.roff + 8 * (p.n_r + i), 1.0);   // dummy rho (column padding): neutral in a product, log = 0
        for (int i = 0; i < 7; i++) if (i != 2) ctl[i] = 0;
        pop();
    }
    __syncthreads();
    int b = __builtin_amdgcn_readfirstlane(ctl[2]);
    while (b >= 0) {
        double *__restrict__ in = p.llr + (int64_t)b * p.n_v;
        for (int v0 = tid; v0 < p.n_v; v0 += 4 * nt) {           // four requests in flight per thread (see ldpc_resident_kernel)
            double rawv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) rawv[u] = in[v0 + u * nt < p.n_v ? v0 + u * nt : v0];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int v = v0 + u * nt;
                if (v >= p.n_v) break;
                const double raw = rawv[u];
                const double x = clip_nan(raw, -500.0, 500.0);
                if (x != raw) {                                  // in-place clip (:186); untouched values are not rewritten
                    if (x == x) {
                        in[v] = x;
                        if (p.clipped) *p.clipped = 1;
                    } else {
                        ctl[3] = 1;                              // NaN
                    }
                }
                const double e0 = exp(x);                        // |x| <= 500: a normal number
                e0row[v] = e0;
                stsd(8 * v, x < 0.0 ? -e0 : e0);                 // out_llrs = llr (:194)
            }
        }
        for (int e = tid; e < p.n_r; e += nt) stsd(p.roff + 8 * e, 1.0);
        __syncthreads();
        int k = 0, back = 0;
        for (; k < p.max_iter; k++) {
            int *flag = &ctl[k & 1];
            for (int c = tid; c < p.n_c; c += nt) check_spa_ratio(p, c, ctl, flag, &ctl[5 + (k & 1)]);
            __syncthreads();
            back = __builtin_amdgcn_readfirstlane(ctl[3]);       // a NaN: log-domain row, whatever the syndrome says
            if (back) break;
            if (!__builtin_amdgcn_readfirstlane(*flag)) break;   // zero syndrome (:205-206): the rho of this pass are never used
            back = 2 * __builtin_amdgcn_readfirstlane(ctl[5 + (k & 1)]) > p.n_c;   // a saturated iteration whose messages WOULD be used
            if (back) break;
            if (tid == 0) {
                ctl[(k + 1) & 1] = 0;
                ctl[5 + ((k + 1) & 1)] = 0;
            }
            for (int v = tid; v < p.n_v; v += nt) var_node_ratio(p, v, e0row[v], in);
            __syncthreads();
        }
        if (back) k = spa_block_log(p, in, ctl, tid, nt);        // wave- and workgroup-uniform
        // Retire: out_llrs = log X (a block that arrived as a codeword: the clipped LLRs themselves, :194; a log-domain block: Q)
        double *__restrict__ orow = p.out + (int64_t)b * p.n_v;
        int8_t *__restrict__ drow = p.dec ? p.dec + (int64_t)b * p.n_v : nullptr;
        for (int v = tid; v < p.n_v; v += nt) {
            const double s0 = ldsd(8 * v);                       // the same thread reloads this entry for the next block
            const double x = back ? s0 : (k == 0 ? in[v] : ratio_llr(fabs(s0)));
            orow[v] = x;
            if (drow) drow[v] = (int8_t)(__builtin_signbit(x) ? 1 : 0);
        }
        if (tid == 0) {
            if (p.iters) p.iters[b] = k;
            if (back)
                for (int i = 0; i < 5; i++) stsd(p.roff + 8 * (p.n_r + i), 1.0);             // column padding: neutral in a PRODUCT again
            for (int i = 0; i < 7; i++) if (i != 2) ctl[i] = 0;
            pop();
        }
        __syncthreads();
        b = __builtin_amdgcn_readfirstlane(ctl[2]);
    }
}

// ---- "fp32-fast" precision mode (cpx_set_precision): the same persistent kernel with float32 state -----------------
// NOT the parity mode.  Q and R are float (half the LDS: four workgroups per compute unit for (1944,1296)), the channel LLRs are
// converted on the way in and the a-posteriori LLRs on the way out; sum-product uses the hardware exp2 / log2 / reciprocal
// (__expf, __logf, __fdividef).  Float32 messages saturate earlier (tanh(m/2) rounds to 1 from |m| ~ 17), so trajectories differ
// from the float64 decoder's: the contract is the decoded word on blocks that converge and the frame error rate, measured in
// tests/test_fp32_fast_gpu.py and DESIGN.md 4.3 -- never the 1e-5 LLR criterion.  Tables: the same layout with 4-byte offsets.
typedef __attribute__((address_space(3))) float lds_f32;
__device__ __forceinline__ float ldsf(int off) { return *reinterpret_cast<lds_f32 *>((unsigned)off); }
__device__ __forceinline__ void stsf(int off, float v) { *reinterpret_cast<lds_f32 *>((unsigned)off) = v; }
__device__ __forceinline__ float clip_nan_f(float v, float lo, float hi) { return (v != v) ? v : fminf(fmaxf(v, lo), hi); }

__device__ __forceinline__ void check_msa_f32(const ResParams &p, int c, int *flag) {
    const int4 *__restrict__ qv = reinterpret_cast<const int4 *>(p.row_q + (int64_t)c * p.cpad);
    const int rb = p.roff + 4 * c * p.rstride, nq = p.cpad >> 2;
    int sx = 0, imin = 0;
    unsigned neg = 0;
    float m1 = __builtin_huge_valf(), m2 = __builtin_huge_valf();
    for (int t = 0; t < nq; t++) {
        const int4 a = qv[t];
        const int off[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = 4 * t + u;
            const float q = ldsf(off[u]);                        // padding: +inf, neutral below
            sx ^= __float_as_int(q);
            const float m = q - ldsf(rb + 4 * j);
            const float av = fabsf(m);
            const bool c1 = av < m1;
            m2 = fminf(m2, c1 ? m1 : av);
            m1 = fminf(m1, av);
            imin = c1 ? j : imin;
            neg |= (m < 0.0f) ? (1u << j) : 0u;
        }
    }
    if (sx < 0) *flag = 1;
    const unsigned negp = (__popc(neg) & 1) ? ~neg : neg;
    const int b1 = __float_as_int(m1);
    for (int j = 0; j < p.rstride; j++) stsf(rb + 4 * j, __int_as_float(b1 | (int)(((negp >> j) & 1u) << 31)));
    stsf(rb + 4 * imin, __int_as_float(__float_as_int(m2) | (int)(((negp >> imin) & 1u) << 31)));
}

__device__ __forceinline__ void check_spa_f32(const ResParams &p, int c, int *flag) {
    const int deg = p.row_deg[c];
    const int4 *__restrict__ qv = reinterpret_cast<const int4 *>(p.row_q + (int64_t)c * p.cpad);
    const int rb = p.roff + 4 * c * p.rstride;
    int sx = 0;
    float prod = 1.0f;
    for (int j0 = 0; __builtin_amdgcn_ballot_w64(j0 < deg) != 0; j0 += 4) {
        const int4 a = qv[j0 >> 2];
        const float q[4] = {ldsf(a.x), ldsf(a.y), ldsf(a.z), ldsf(a.w)};
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + u;
            if (j < deg) {
                sx ^= __float_as_int(q[u]);
                const float m = q[u] - ldsf(rb + 4 * j);
                const float e = __expf(-fabsf(m));
                const float t = __builtin_copysignf(__fdividef(1.0f - e, 1.0f + e), m);     // tanh(m / 2)
                prod *= t;
                stsf(rb + 4 * j, t);
            }
        }
    }
    if (sx < 0) *flag = 1;
    for (int j = 0; __builtin_amdgcn_ballot_w64(j < deg) != 0; j++) {
        if (j < deg) {
            float x = __fdividef(1.0f, ldsf(rb + 4 * j)) * prod;
            x = clip_nan_f(x, -1.0f, 1.0f);
            x = __logf(__fdividef(1.0f + x, 1.0f - x));                                     // 2 atanh(x)
            stsf(rb + 4 * j, clip_nan_f(x, -500.0f, 500.0f));
        }
    }
}

__device__ __forceinline__ void var_node_f32(const ResParams &p, int v, const double *__restrict__ lrow) {
    const int trips = p.vgrp[__builtin_amdgcn_readfirstlane(v) >> 6];
    const int4 *__restrict__ rf = reinterpret_cast<const int4 *>(p.col_r + (int64_t)v * p.vpad);
    const float l = (float)lrow[v];
    float msum = 0.0f;
    for (int t = 0; t < trips; t++) {
        const int4 a = rf[t];
        const float r0 = ldsf(a.x), r1 = ldsf(a.y), r2 = ldsf(a.z), r3 = ldsf(a.w);
        msum += r0; msum += r1; msum += r2; msum += r3;
    }
    stsf(4 * v, msum + l);
}

template <int ALG>
__global__ __launch_bounds__(1024, 8) void ldpc_resident_f32_kernel(ResParams p) {
    extern __shared__ __align__(16) char lds[];
    if ((unsigned)(uintptr_t)lds != 0u) __builtin_trap();
    int *ctl = reinterpret_cast<int *>(lds + p.ctl_off);
    const int tid = threadIdx.x, nt = blockDim.x;
    auto pop = [&]() {                                           // thread 0 only (control flow: see ldpc_resident_kernel)
        const int t = atomicAdd(p.queue, 1);
        ctl[2] = t < p.B ? t : -1;
    };
    if (tid == 0) {
        stsf(4 * p.n_v, __builtin_huge_valf());
        stsf(p.roff + 4 * p.n_r, 0.0f);
        ctl[0] = 0; ctl[1] = 0;
        ctl[3] = 0;
        pop();
    }
    __syncthreads();
    int b = __builtin_amdgcn_readfirstlane(ctl[2]);
    while (b >= 0) {
        double *__restrict__ in = p.llr + (int64_t)b * p.n_v;
        for (int v0 = tid; v0 < p.n_v; v0 += 4 * nt) {           // four requests in flight per thread (see ldpc_resident_kernel)
            double rawv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) rawv[u] = in[v0 + u * nt < p.n_v ? v0 + u * nt : v0];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int v = v0 + u * nt;
                if (v >= p.n_v) break;
                const double raw = rawv[u];
                const double x = clip_nan(raw, -500.0, 500.0);
                if (x != raw) {                                  // in-place clip (:186), as in the parity kernel
                    if (x == x) {
                        in[v] = x;
                        if (p.clipped) *p.clipped = 1;
                    } else {
                        ctl[3] = 1;
                    }
                }
                stsf(4 * v, (float)x);
            }
        }
        for (int e = tid; e < p.n_r; e += nt) stsf(p.roff + 4 * e, 0.0f);
        __syncthreads();
        int k = 0;
        for (; k < p.max_iter; k++) {
            int *flag = &ctl[k & 1];
            for (int c = tid; c < p.n_c; c += nt) {
                if (ALG == CPX_LDPC_MSA) check_msa_f32(p, c, flag);
                else check_spa_f32(p, c, flag);
            }
            __syncthreads();
            if (!__builtin_amdgcn_readfirstlane(*flag)) break;
            if (tid == 0) ctl[(k + 1) & 1] = 0;
            for (int v = tid; v < p.n_v; v += nt) var_node_f32(p, v, in);
            __syncthreads();
        }
        // Retire: the a-posteriori LLRs (and, for block-major outputs, their sign bits) as one contiguous row -- the layout the
        // reference's own results have in memory (ldpc.py:251-253 reshapes with order='F': one block per column of an F-ordered
        // array IS one block per contiguous row).  Not waited for: the next block's loads queue behind these stores.
        // A block that was a codeword on arrival (k == 0: no variable pass ran) returns its clipped input LLRs UNCHANGED like the
        // reference (out_llrs = llr_vec.copy(), ldpc.py:194) -- the float64 input, not its float32 image in LDS.
        double *__restrict__ orow = p.out + (int64_t)b * p.n_v;
        const bool untouched = k == 0;
        if (p.dec) {
            int8_t *__restrict__ drow = p.dec + (int64_t)b * p.n_v;
            for (int v = tid; v < p.n_v; v += nt) {
                const double x = untouched ? in[v] : (double)ldsf(4 * v);       // the same thread reloads these entries for the next block
                orow[v] = x;                                     // (:247)
                drow[v] = (int8_t)(__builtin_signbit(x) ? 1 : 0);    // (:248)
            }
        } else {
            for (int v = tid; v < p.n_v; v += nt) orow[v] = untouched ? in[v] : (double)ldsf(4 * v);
        }
        if (tid == 0) {
            if (p.iters) p.iters[b] = k;
            if (p.nanflags) p.nanflags[b] = (uint8_t)ctl[3];
            ctl[3] = 0;
            ctl[0] = 0; ctl[1] = 0;                              // (a thread that has yet to read a flag of this block reads the 0 that sent the others here)
            pop();
        }
        __syncthreads();
        b = __builtin_amdgcn_readfirstlane(ctl[2]);
    }
}

// staging [B][n_v] -> out_llrs [n_v][B], dec_word [n_v][B] (:247-253)
__global__ __launch_bounds__(256) void ldpc_unstage_kernel(const double *__restrict__ stage, int64_t B, int n_v,
                                                           double *__restrict__ out, int8_t *__restrict__ dec) {
    __shared__ double ts[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t b0 = (int64_t)blockIdx.x * 64;
    const int v0 = blockIdx.y * 64;
    for (int r = ty; r < 64; r += 4) {                            // rows = blocks, columns = variables
        const int64_t b = b0 + r;
        const int v = v0 + tx;
        ts[r][tx] = (b < B && v < n_v) ? stage[b * n_v + v] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {                            // rows = variables, columns = blocks
        const int v = v0 + r;
        const int64_t b = b0 + tx;
        if (v < n_v && b < B) {
            const double x = ts[tx][r];
            out[(int64_t)v * B + b] = x;
            dec[(int64_t)v * B + b] = (int8_t)(__builtin_signbit(x) ? 1 : 0);
        }
    }
}

std::atomic<int> g_ldpc_path{-1};                                 // 0 auto, 1 tiled, 2 resident (strict), 3 resident (strict), sum-product by the log-domain row
int parse_ldpc_path(const char *m) {
    if (!m || !m[0] || strcmp(m, "auto") == 0) return 0;
    if (strcmp(m, "tiled") == 0) return 1;
    if (strcmp(m, "resident") == 0) return 2;
    if (strcmp(m, "resident-log") == 0) return 3;
    return -2;
}
int ldpc_path() {
    int v = g_ldpc_path.load(std::memory_order_relaxed);
    if (v < 0) {
        static std::once_flag once;
        std::call_once(once, [] {
            const int e = parse_ldpc_path(getenv("CPX_LDPC_PATH"));
            g_ldpc_path.store(e < 0 ? 0 : e, std::memory_order_relaxed);
        });
        v = g_ldpc_path.load(std::memory_order_relaxed);
    }
    return v;
}

template <int ALG, int CQ>
int launch_resident(const ResParams &p, int grid, int threads, size_t lds, hipStream_t st) {
    static bool raised[64] = {};                                  // > 64 KiB of dynamic LDS is opt-in, once per kernel and device
    static std::mutex raised_mu;
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lk(raised_mu);
        if (dev >= 0 && dev < 64 && !raised[dev]) {
            CPX_HIP(hipFuncSetAttribute((const void *)ldpc_resident_kernel<ALG, CQ>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)LDS_BYTES));
            raised[dev] = true;
        }
    }
    hipLaunchKernelGGL((ldpc_resident_kernel<ALG, CQ>), dim3((unsigned)grid), dim3((unsigned)threads), lds, st, p);
    return CPX_OK;
}

int launch_ratio(const ResParams &p, int grid, int threads, size_t lds, hipStream_t st) {
    static bool raised[64] = {};
    static std::mutex raised_mu;
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lk(raised_mu);
        if (dev >= 0 && dev < 64 && !raised[dev]) {
            CPX_HIP(hipFuncSetAttribute((const void *)ldpc_resident_ratio_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
            raised[dev] = true;
        }
    }
    hipLaunchKernelGGL(ldpc_resident_ratio_kernel, dim3((unsigned)grid), dim3((unsigned)threads), lds, st, p);
    return CPX_OK;
}

template <int ALG>
int launch_resident_f32(const ResParams &p, int grid, int threads, size_t lds, hipStream_t st) {
    hipLaunchKernelGGL((ldpc_resident_f32_kernel<ALG>), dim3((unsigned)grid), dim3((unsigned)threads), lds, st, p);   // < 64 KB of LDS
    return CPX_OK;
}

int res_roff(int n_v) { return ((n_v + 2) & ~1) * 8; }            // Q[n_v] + the dummy slot, R 16-byte aligned
int res_rstride(const cpx_ldpc *c) { return c->max_cdeg | 1; }
size_t res_lds_bytes_f32(const cpx_ldpc *c) { return (size_t)(res_roff(c->n_v) / 2) + 4 * ((size_t)c->n_c * (c->max_cdeg | 1) + 1 + 4) + 64; }
size_t res_lds_bytes(const cpx_ldpc *c) { return (size_t)res_roff(c->n_v) + 8 * ((size_t)c->n_c * res_rstride(c) + 1 + 4) + 64; }
}  // namespace

namespace cpx {

bool ldpc_spa_exact() {
    static const bool v = [] { const char *e = getenv("CPX_LDPC_SPA"); return e && strcmp(e, "exact") == 0; }();
    return v;
}
// CPX_LDPC_SPA=log: the resident path keeps the log-domain one-division row (the round-3 kernel) instead of the ratio-domain kernel (A/B runs)
static bool ldpc_spa_log() {
    static const bool v = [] { const char *e = getenv("CPX_LDPC_SPA"); return e && strcmp(e, "log") == 0; }();
    return v;
}

// Offset tables of the resident path, built once per handle from the blob's tables (host pointers).
int ldpc_resident_tables(cpx_ldpc *c, const int32_t *row_ptr, const int32_t *row_pad, const int32_t *col_ptr,
                         const int32_t *col_pad_cj) {
    if (res_lds_bytes(c) > LDS_BYTES) return CPX_OK;              // does not fit: the handle only serves the tiled path
    if (c->max_cdeg > 32) return CPX_OK;                          // 5-bit row positions: such a code takes ldpc_exact_kernel (ldpc.hip)
    const int n_v = c->n_v, n_c = c->n_c, cpad = c->cpad, vpad = c->vpad;
    const int roff = res_roff(n_v), rs = res_rstride(c);
    std::vector<int32_t> dg((size_t)n_c), rq((size_t)n_c * cpad + 16, 8 * n_v),
        cr((size_t)n_v * vpad + 16, roff + 8 * n_c * rs), vg((size_t)(n_v + 63) / 64, 0);
    for (int k = 0; k < n_c; k++) {
        const int d = row_ptr[k + 1] - row_ptr[k];
        dg[k] = d;
        for (int j = 0; j < d; j++) rq[(size_t)k * cpad + j] = 8 * row_pad[(size_t)k * cpad + j];
    }
    for (int v = 0; v < n_v; v++) {
        const int d = col_ptr[v + 1] - col_ptr[v];
        vg[v >> 6] = std::max(vg[v >> 6], (d + 3) / 4);
        for (int q = 0; q < d; q++) {
            const int cj = col_pad_cj[(size_t)v * vpad + q];      // (check << 5) | position of the edge in its row
            cr[(size_t)v * vpad + q] = roff + 8 * ((cj >> 5) * rs + (cj & 31));
        }
    }
    // the float32 mode: same tables with 4-byte offsets (R starts at roff / 2)
    std::vector<int32_t> rq32(rq.size()), cr32(cr.size());
    for (size_t i = 0; i < rq.size(); i++) rq32[i] = rq[i] / 2;
    for (size_t i = 0; i < cr.size(); i++) cr32[i] = roff / 2 + (cr[i] - roff) / 2;
    struct Up { int32_t **dst; std::vector<int32_t> *src; } ups[] = {{&c->d_res_row_q32, &rq32}, {&c->d_res_col_r32, &cr32}, {&c->d_res_row_deg, &dg}, {&c->d_res_row_q, &rq},
                                                                     {&c->d_res_col_r, &cr}, {&c->d_res_vgrp, &vg}};
    for (auto &u : ups) {
        hipError_t e1 = hipMalloc((void **)u.dst, sizeof(int32_t) * u.src->size());
        hipError_t e2 = e1 == hipSuccess ? hipMemcpy(*u.dst, u.src->data(), sizeof(int32_t) * u.src->size(), hipMemcpyHostToDevice) : e1;
        if (e2 != hipSuccess) { set_error("cpx_ldpc_create: device upload failed: %s", hipGetErrorString(e2)); return CPX_EHIP; }
    }
    return CPX_OK;
}

void ldpc_resident_free(cpx_ldpc *c) {
    (void)hipFree(c->d_res_row_deg); (void)hipFree(c->d_res_row_q); (void)hipFree(c->d_res_col_r); (void)hipFree(c->d_res_vgrp); (void)hipFree(c->d_res_row_q32); (void)hipFree(c->d_res_col_r32);
}

int ldpc_forced_path() { return ldpc_path(); }

bool ldpc_resident_path(const cpx_ldpc *c, double *d_llr, int64_t B, int alg, int n_iters, int8_t *d_dec, double *d_out,
                        int block_major, int32_t *d_iters, int *d_clipped, uint8_t *nanflags, hipStream_t st, int *rc) {
    *rc = CPX_OK;
    const int mode = ldpc_path();
    auto reject = [&](const char *why) {
        if (mode >= 2) { set_error("ldpc: resident path forced but not applicable: %s", why); *rc = CPX_EINVAL; return true; }
        return false;
    };
    if (mode == 1) return false;
    if (n_iters < 1) return reject("n_iters == 0");
    if (B >= (1ll << 30)) return reject("batch too large");
    if (!c->d_res_row_q) return reject("decoder state of one block exceeds the LDS of a compute unit");
    const bool f32 = precision_fast() && res_lds_bytes_f32(c) <= 64 * 1024;   // "fp32-fast": float32 state (ldpc_resident_f32_kernel)
    const size_t lds = f32 ? res_lds_bytes_f32(c) : res_lds_bytes(c);
    // workgroup size.  float64: the check pass in one round (two for > 1024 checks) -- (1944,1296): 704 threads, the variable
    // pass takes three rounds; measured 2.51 ms at 704, 2.75 at 512, 3.38 at 1024 (scripts/micro/ldpc_knobs.py).  float32: 512.
    const int rounds = (c->n_c + 1023) / 1024;
    int threads = f32 ? std::min(512, std::max(64, (c->n_c + 63) / 64 * 64))
                      : std::min(1024, std::max(64, ((c->n_c + rounds - 1) / rounds + 63) / 64 * 64));
    if (const char *e = getenv("CPX_LDPC_THREADS")) {             // experiment knob
        const int t = atoi(e);
        if (!f32 && t >= 64 && t <= 1024 && t % 64 == 0) threads = t;
    }
    // Outputs.  Block-major (one block per row, the memory layout of the reference's own results): retired blocks go straight
    // to the caller's arrays.  [n_v][B]: they go to a staging buffer and ldpc_unstage_kernel transposes it.
    // persistent grid: as many workgroups as fit a compute unit (LDS, 2048 threads, 16 workgroups), no more than blocks
    const int per_cu = std::max(1, std::min({(int)(LDS_BYTES / lds), 2048 / threads, 16}));
    const int grid = (int)std::min<int64_t>((int64_t)device_cus() * per_cu, B);
    const bool ratio = !f32 && alg == CPX_LDPC_SPA && !ldpc_spa_exact() && !ldpc_spa_log() && mode != 3;   // ldpc_resident_ratio_kernel
    char *slab = nullptr;
    const size_t sz_stage = block_major ? 0 : (sizeof(double) * (size_t)(B * c->n_v) + 255) & ~(size_t)255;
    const size_t sz_e0 = ratio ? sizeof(double) * (size_t)grid * (size_t)c->n_v : 0;
    if ((*rc = workspace(st, 0, sz_stage + 256 + sz_e0, (void **)&slab))) return true;
    ResParams p;
    p.llr = d_llr; p.iters = d_iters; p.queue = (int *)(slab + sz_stage); p.clipped = d_clipped; p.nanflags = nanflags;
    p.e0 = (double *)(slab + sz_stage + 256);
    p.out = block_major ? d_out : (double *)slab;
    p.dec = block_major ? d_dec : nullptr;
    p.spa_exact = (!f32 && ldpc_spa_exact()) ? 1 : 0;
    p.row_deg = c->d_res_row_deg; p.vgrp = c->d_res_vgrp;
    p.row_q = f32 ? c->d_res_row_q32 : c->d_res_row_q;
    p.col_r = f32 ? c->d_res_col_r32 : c->d_res_col_r;
    p.B = B; p.rstride = res_rstride(c); p.n_r = c->n_c * p.rstride; p.n_v = c->n_v; p.n_c = c->n_c; p.cpad = c->cpad; p.vpad = c->vpad;
    p.max_iter = n_iters;
    p.roff = f32 ? res_roff(c->n_v) / 2 : res_roff(c->n_v);
    p.ctl_off = (int)(lds - 64);
    if (hipMemsetAsync(p.queue, 0, sizeof(int), st) != hipSuccess) { set_error("ldpc: hipMemsetAsync failed"); *rc = CPX_EHIP; return true; }
    const int cq = c->cpad / 4;
    int lrc;
    if (ratio) {
        lrc = launch_ratio(p, grid, threads, lds, st);
    } else if (f32) lrc = alg == CPX_LDPC_SPA ? launch_resident_f32<CPX_LDPC_SPA>(p, grid, threads, lds, st)
                                       : launch_resident_f32<CPX_LDPC_MSA>(p, grid, threads, lds, st);
    else if (alg == CPX_LDPC_SPA) lrc = launch_resident<CPX_LDPC_SPA, 0>(p, grid, threads, lds, st);
    else if (cq == 1) lrc = launch_resident<CPX_LDPC_MSA, 1>(p, grid, threads, lds, st);
    else if (cq == 2) lrc = launch_resident<CPX_LDPC_MSA, 2>(p, grid, threads, lds, st);
    else if (cq == 3) lrc = launch_resident<CPX_LDPC_MSA, 3>(p, grid, threads, lds, st);
    else if (cq == 4) lrc = launch_resident<CPX_LDPC_MSA, 4>(p, grid, threads, lds, st);
    else lrc = launch_resident<CPX_LDPC_MSA, 0>(p, grid, threads, lds, st);
    if (lrc) { *rc = lrc; return true; }
    if (!block_major)
        hipLaunchKernelGGL(ldpc_unstage_kernel, dim3((unsigned)((B + 63) / 64), (unsigned)((c->n_v + 63) / 64)), dim3(256), 0, st,
                           p.out, B, c->n_v, d_out, d_dec);
    if (hipGetLastError() != hipSuccess) { set_error("ldpc (resident path): launch failed"); *rc = CPX_EHIP; }
    if (f32) note_kernel("ldpc_resident_f32_kernel<%s> threads=%d workgroups/CU=%d", alg == CPX_LDPC_MSA ? "MSA" : "SPA", threads, per_cu);
    else if (ratio) note_kernel("ldpc_resident_kernel<SPA,ratio> threads=%d workgroups/CU=%d", threads, per_cu);
    else note_kernel("ldpc_resident_kernel<%s,%d> threads=%d workgroups/CU=%d", alg == CPX_LDPC_MSA ? "MSA" : "SPA",
                     alg == CPX_LDPC_MSA && cq <= 4 ? cq : 0, threads, per_cu);
    return true;
}

}  // namespace cpx

extern "C" int cpx_ldpc_set_path(const char *mode) {
    const int v = parse_ldpc_path(mode);
    if (v < 0) {
        cpx::set_error("cpx_ldpc_set_path: unknown mode '%s' (auto | tiled | resident | resident-log)", mode);
        return CPX_EINVAL;
    }
    g_ldpc_path.store(v, std::memory_order_relaxed);
    return CPX_OK;
}
