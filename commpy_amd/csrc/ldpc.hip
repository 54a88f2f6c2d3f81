// LDPC belief propagation (flooding schedule, SPA and min-sum) for gfx950.  Replaces the body of
//   ldpc_bp_decode  (/root/reference/commpy/channelcoding/ldpc.py:144-254)
// for a batch of B independent blocks (the reference's sequential `for i_start` loop, ldpc.py:197).
//
// Formulation.  The reference keeps ONE message per edge and overwrites it twice per iteration
// (check pass :209-238, variable pass :243-245).  Its variable pass writes
//     M[e] = M[e] * -1 + 1.0 * (msg_sum[v] + llr[v])                      (:244-245)
// and (msg_sum + llr) is exactly the a-posteriori LLR it also stores in out_llrs (:247).  So the engine keeps
//     R [e]  = check->variable message (the state after the check pass)
//     Q [v]  = msg_sum + llr            (the a-posteriori LLR, also the decision variable)
// and the check pass recomputes the variable->check message as R*(-1) + 1.0*Q -- the same two float64
// operations on the same operands, hence bit-identical -- instead of reading it back from memory.
// HBM traffic per executed iteration per block falls from (4E + 2n)*8 B to about (3E + 3n)*8 B for SPA
// (R read+written by the check pass, read by the variable pass; Q, llr: n each; Q re-reads hit L2).
// Min-sum goes further: the messages a check sends are +-(one of two magnitudes), so its row is kept as a
// 3-word record (min1, min2, position/sign bits -- see MsaRec) instead of deg_c float64: about
// (9 n_c + 3n)*8 B per iteration, 2.8x less than the reference formulation for the (1944,1296) code, still
// bit-identical (the record stores the generating values, not an approximation).
// The parity of sign(Q) over a check's row is the syndrome bit (:203-206), so the early-termination test
// rides in the check pass: a pass marks the block "unsatisfied" when it sees an odd row, and the variable pass
// only commits Q for marked blocks.  A block whose syndrome is zero therefore executes one speculative check
// pass whose R is never used again, and then freezes -- the reference's `break`.
//
// Data layout in HBM: tiles of 64 blocks (one wavefront), tile-major, block index fastest:
//   R [tile][E][64] (SPA) or [tile][n_c][3][64] (min-sum records), Q [tile][n_v][64], L [tile][n_v][64]
//   float64;  state/orig [slot] int32
// Every access is a fully coalesced 512-byte row.  A slot's state is the stamp of the last iteration that
// found it unsatisfied (+1), -1 for padding/retired slots; orig is the block index in the caller's arrays.
//
// Launch geometry: persistent grids sized to what is resident (occupancy x CUs, a multiple of 8 workgroups),
// wave = one (tile, check) or (tile, variable) item; node tables are padded so that every scalar load of an
// item depends on the node index only, and the min-sum check pass prefetches the next item's operands.  Workgroups are dealt round-robin to the 8 XCDs, so workgroup id mod 8 selects the tiles it serves:
// all checks of a tile run on one XCD and the tile's 1 MB of Q (gathered deg_v times) stays in that XCD's L2.
//
// Compaction.  Frozen blocks would keep occupying lanes of half-empty wavefronts (partial cache lines cost
// full transactions).  After every iteration a single-workgroup scan counts the unsatisfied slots; when they are
// at most half of the live slots, a move pass copies their R/Q/L columns to the front of a second set of
// buffers and *retires* the frozen ones (out_llrs / dec_word written at their original block index).  All of it
// is decided and applied on the device (control block in HBM, no host synchronisation).
#include "cpx_internal.h"
#include "cpx_math.h"
#include "ldpc_dev.h"

#include <algorithm>

using namespace cpx;

namespace {

constexpr int LB = 256;     // threads per workgroup = 4 wavefronts
constexpr int MAXDEG = 32;  // max check degree held in registers

struct Ctl { int n_slots, buf, do_move, n_new, prev_active; };

struct Bufs {
    double *R[2], *Q[2], *L[2];
    int32_t *state[2], *orig[2];
    int32_t *dst;
    Ctl *ctl;
};

// view of the control block for compute passes: a pending move is already in effect for them
__device__ __forceinline__ void effective(const Ctl *c, int &n_slots, int &buf) {
    const int mv = c->do_move;
    n_slots = mv ? c->n_new : c->n_slots;
    buf = c->buf ^ mv;
}

// L = Q = clip(llr) tile-major; llr clipped in place (ldpc.py:186); slot table; control block
__global__ __launch_bounds__(LB) void ldpc_init_kernel(double *__restrict__ llr, int64_t B, int n_v, Bufs bf,
                                                       int32_t *__restrict__ iters, int *__restrict__ clipped,
                                                       uint8_t *__restrict__ nanflags /* min-sum: [B], zeroed; else null */) {
    __shared__ double ts[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;      // 64 x 4
    const int64_t tile = blockIdx.x;
    const int v0 = blockIdx.y * 64;
    for (int r = ty; r < 64; r += 4) {                            // rows = blocks, columns = variables
        const int64_t b = tile * 64 + r;
        const int v = v0 + tx;
        double x = 0.0;
        if (b < B && v < n_v) {
            const double raw = llr[b * n_v + v];
            x = clip_nan(raw, -500.0, 500.0);
            if (x != raw) {                                       // in-place clip (:186); untouched values are not rewritten
                if (x == x) {
                    llr[b * n_v + v] = x;
                    if (clipped) *clipped = 1;                    // lets the host-buffer entry point skip the copy back
                } else if (nanflags) {
                    nanflags[b] = 1;                              // min-sum: the block is decoded again, NaN-exact (ldpc_exact_kernel<false>)
                }
            }
        }
        ts[r][tx] = x;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {                            // rows = variables, columns = blocks
        const int v = v0 + r;
        if (v < n_v) {
            const double x = ts[tx][r];
            const int64_t i = (tile * n_v + v) * 64 + tx;
            bf.L[0][i] = x;
            bf.Q[0][i] = x;                                       // out_llrs = llr (:194)
        }
    }
    if (blockIdx.y == 0 && threadIdx.x < 64) {
        const int64_t slot = tile * 64 + threadIdx.x;
        bf.state[0][slot] = slot < B ? 0 : -1;
        bf.orig[0][slot] = (int32_t)slot;
        if (slot < B && iters) iters[slot] = 0;
        if (slot == 0) { bf.ctl->n_slots = (int)(gridDim.x * 64); bf.ctl->buf = 0; bf.ctl->do_move = 0; bf.ctl->n_new = 0; bf.ctl->prev_active = (int)(gridDim.x * 64); }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Work distribution of the compute passes: wave w of workgroup `blockIdx.x` serves group member g*4+w of
// (tile, group g); tiles are dealt by workgroup id mod 8 so that one XCD owns a tile.
#define CPX_ITEM_LOOP(N_GROUPS)                                                                      \
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;                                         \
    const int xcd = blockIdx.x & 7, jstride = gridDim.x >> 3;                                        \
    for (int idx = blockIdx.x >> 3;; idx += jstride)

// ---- sum-product (:209-227): R keeps one float64 per edge --------------------------------------------------
// The row of ldpc_dev.h: one division per edge, rows near saturation by the exact-order sequence -- the same operations in
// the same order as check_spa of the LDS-resident path (results are bit-identical between the two paths).
template <int DEG>
__device__ __forceinline__ void cn_spa_row(double *__restrict__ Rrow, const double *__restrict__ Qt,
                                           const int32_t *__restrict__ ev, int deg, int k, int32_t *st, int spa_exact) {
    // DEG > 0: exact degree, fully unrolled, row in registers;  DEG == 0: any degree (13 .. 32 edges), the row's e values parked in R.
    // Either way every operand of the row -- of a chunk of eight edges in the rolled form -- is requested before the first one is
    // used (round 5: written edge by edge the compiler waited for each load in turn -- 22 memory latencies per 11-edge row, 13 us per
    // row and wavefront, the pass at 1.7 TB/s instead of 4.3).
    constexpr int CH = 8;
    int sx = 0;
    double U = 1.0, W = 1.0, emax = 0.0;
    auto edge_in = [&](double q, double r) -> double {           // r: the edge's R of the previous iteration (not looked at when k == 0)
        sx ^= __double2hiint(q);                                 // dec_word = out_llrs < 0 (:193, :248)
        double m = 1.0 * q;                                      // message_matrix = H.multiply(llr) (:199)
        if (k > 0) {
            m = r * -1.0;                                        // data *= -1 (:244)
            m += 1.0 * q;                                        // data += H.multiply(msg_sum + llr).data (:245)
        }
        double se, u, w;
        spa_in(m, se, u, w);                                     // e = exp(-|m|): tanh(m / 2) = u / w (:210-211)
        U *= u; W *= w;                                          // row product (reference: exp2(sum(log2)) :217-219)
        emax = fmax(emax, fabs(se));
        return se;
    };
    if constexpr (DEG > 0) {
        double v[DEG], qv[DEG], rv[DEG];
#pragma unroll
        for (int j = 0; j < DEG; j++) qv[j] = Qt[(int64_t)ev[j] * 64];
#pragma unroll
        for (int j = 0; j < DEG; j++) rv[j] = 0.0;
        if (k > 0) {
#pragma unroll
            for (int j = 0; j < DEG; j++) rv[j] = ntload(&Rrow[(int64_t)j * 64]);
        }
#pragma unroll
        for (int j = 0; j < DEG; j++) v[j] = edge_in(qv[j], rv[j]);
        if (sx < 0) *st = k + 1;                                 // odd row: iteration k is executed (:205)
        if (!(spa_exact || spa_row_near(U, W, emax))) {
#pragma unroll
            for (int j = 0; j < DEG; j++) ntstore(spa_out_fast(U, W, v[j]), &Rrow[(int64_t)j * 64]);
        } else {                                                 // near saturation: the exact-order sequence (ldpc_dev.h)
            double prod = 1.0;
#pragma unroll
            for (int j = 0; j < DEG; j++) prod *= spa_exact_t(v[j]);
#pragma unroll
            for (int j = 0; j < DEG; j++) ntstore(spa_out_exact(spa_exact_t(v[j]), prod), &Rrow[(int64_t)j * 64]);
        }
    } else {
        // (an index past the row's end is clamped to its last edge: a valid address, loaded again and not used)
        for (int j0 = 0; j0 < deg; j0 += CH) {
            double qq[CH], rr[CH];
#pragma unroll
            for (int u = 0; u < CH; u++) qq[u] = Qt[(int64_t)ev[j0 + u < deg ? j0 + u : deg - 1] * 64];
#pragma unroll
            for (int u = 0; u < CH; u++) rr[u] = 0.0;
            if (k > 0) {
#pragma unroll
                for (int u = 0; u < CH; u++) rr[u] = ntload(&Rrow[(int64_t)(j0 + u < deg ? j0 + u : deg - 1) * 64]);
            }
#pragma unroll
            for (int u = 0; u < CH; u++)
                if (j0 + u < deg) Rrow[(int64_t)(j0 + u) * 64] = edge_in(qq[u], rr[u]);
        }
        if (sx < 0) *st = k + 1;                                 // odd row: iteration k is executed (:205)
        const bool near = spa_exact || spa_row_near(U, W, emax);
        double prod = 1.0;
        if (near) {
            for (int j0 = 0; j0 < deg; j0 += CH) {
                double se[CH];
#pragma unroll
                for (int u = 0; u < CH; u++) se[u] = Rrow[(int64_t)(j0 + u < deg ? j0 + u : deg - 1) * 64];
#pragma unroll
                for (int u = 0; u < CH; u++)
                    if (j0 + u < deg) prod *= spa_exact_t(se[u]);
            }
        }
        for (int j0 = 0; j0 < deg; j0 += CH) {
            double se[CH];
#pragma unroll
            for (int u = 0; u < CH; u++) se[u] = Rrow[(int64_t)(j0 + u < deg ? j0 + u : deg - 1) * 64];
            if (!near) {
#pragma unroll
                for (int u = 0; u < CH; u++)
                    if (j0 + u < deg) ntstore(spa_out_fast(U, W, se[u]), &Rrow[(int64_t)(j0 + u) * 64]);
            } else {                                             // near saturation: the exact-order sequence (ldpc_dev.h)
#pragma unroll
                for (int u = 0; u < CH; u++)
                    if (j0 + u < deg) ntstore(spa_out_exact(spa_exact_t(se[u]), prod), &Rrow[(int64_t)(j0 + u) * 64]);
            }
        }
    }
}

// ---- min-sum (:229-238): a row is kept as a record (MsaRec, ldpc_dev.h) -------------------------------------
template <int DEG>
__device__ __forceinline__ void cn_msa_row(const MsaRec &o, double *__restrict__ rec, const double *__restrict__ Qt,
                                           const int32_t *__restrict__ ev, int deg, int k, int32_t *st) {
    int sx = 0, imin = 0;
    unsigned neg = 0;
    double m1 = __builtin_huge_val(), m2 = __builtin_huge_val();
#define CPX_MSA_EDGE(j, QV)                                                                           \
    {                                                                                                 \
        const double q = (QV);                                                                        \
        sx ^= __double2hiint(q);                             /* dec_word = out_llrs < 0 (:193, :248) */ \
        double m = q;                                        /* 1.0 * llr (:199) */                    \
        if (k > 0) m = msa_edge(o, (j), 1) + q;              /* data * -1 + 1.0 * (msg_sum + llr) (:244-245) */ \
        const double a = fabs(m);                                                                     \
        const bool c1 = a < m1;                                                                       \
        m2 = min_f64(m2, c1 ? m1 : a);                                                                \
        m1 = min_f64(m1, a);                                                                          \
        imin = c1 ? (j) : imin;                                                                       \
        neg |= (m < 0.0) ? (1u << (j)) : 0u;                                                          \
    }
    if (DEG > 0) {
#pragma unroll
        for (int j = 0; j < DEG; j++) CPX_MSA_EDGE(j, Qt[(int64_t)ev[j] * 64])
    } else {
        // rolled rows (13 .. 32 edges): eight Q rows requested at a time (an index past the end is clamped to the last edge)
        for (int j0 = 0; j0 < deg; j0 += 8) {
            double qq[8];
#pragma unroll
            for (int u = 0; u < 8; u++) qq[u] = Qt[(int64_t)ev[j0 + u < deg ? j0 + u : deg - 1] * 64];
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (j0 + u < deg) CPX_MSA_EDGE(j0 + u, qq[u])
        }
    }
#undef CPX_MSA_EDGE
    if (sx < 0) *st = k + 1;                                     // odd row: iteration k is executed (:205)
    const int lo = imin | ((__popc(neg) & 1) << 8);
    ntstore(m1, &rec[0]);
    ntstore(m2, &rec[64]);
    ntstore(__hiloint2double((int)neg, lo), &rec[128]);
}

#define CPX_DEG_SWITCH(FN, ...)                                                                      \
    switch (deg) {                                                                                   \
    case 2: FN<2>(__VA_ARGS__); break;   case 3: FN<3>(__VA_ARGS__); break;                          \
    case 4: FN<4>(__VA_ARGS__); break;   case 5: FN<5>(__VA_ARGS__); break;                          \
    case 6: FN<6>(__VA_ARGS__); break;   case 7: FN<7>(__VA_ARGS__); break;                          \
    case 8: FN<8>(__VA_ARGS__); break;   case 9: FN<9>(__VA_ARGS__); break;                          \
    case 10: FN<10>(__VA_ARGS__); break; case 11: FN<11>(__VA_ARGS__); break;                        \
    case 12: FN<12>(__VA_ARGS__); break;                                                             \
    default: FN<0>(__VA_ARGS__); break;                                                              \
    }

// Sum-product check pass of iteration k: syndrome bit + check-node update.  RR = E rows of R per tile.
__global__ __launch_bounds__(LB) void ldpc_cn_spa_kernel(Bufs bf, int n_v, int n_c, int64_t RR,
                                                         const int32_t *__restrict__ row_ptr,
                                                         const int32_t *__restrict__ row_pad, int cpad, int k, int spa_exact) {
    int n_slots, buf;
    effective(bf.ctl, n_slots, buf);
    const int n_tiles = n_slots >> 6;
    double *__restrict__ R = bf.R[buf];
    const double *__restrict__ Q = bf.Q[buf];
    int32_t *__restrict__ state = bf.state[buf];
    const int nG = (n_c + 3) >> 2;
    CPX_ITEM_LOOP(nG) {
        const int tl = idx / nG, cg = idx - tl * nG;
        const int64_t tile = (int64_t)tl * 8 + xcd;
        if (tile >= n_tiles) break;
        const int c = cg * 4 + w;
        if (c >= n_c) continue;
        // everything whose address depends on (tile, c) only is requested before the first wait
        const int64_t slot = tile * 64 + lane;
        const int st = state[slot];
        const int e0 = row_ptr[c];
        const int deg = row_ptr[c + 1] - e0;
        const int32_t *__restrict__ ev = row_pad + (int64_t)c * cpad;
        const double *__restrict__ Qt = Q + tile * n_v * 64 + lane;
        if (st < k) continue;                                     // frozen, retired or padding
        double *__restrict__ Rrow = R + (tile * RR + e0) * 64 + lane;
        CPX_DEG_SWITCH(cn_spa_row, Rrow, Qt, ev, deg, k, &state[slot], spa_exact)
    }
}

// Sum-product column sum in increasing check order (message_matrix.sum(0), :243); refs[q] = edge id.
// DEG > 0: exact degree, all loads in flight; DEG == 0: any degree, four at a time.
template <int DEG>
__device__ __forceinline__ double vn_sum(const double *__restrict__ Rt, const int32_t *__restrict__ refs, int deg) {
    constexpr int CH = DEG > 0 ? DEG : 4;
    double msum = 0.0;
    for (int q0 = 0; q0 < (DEG > 0 ? 1 : deg); q0 += CH) {
        double r[CH];
#pragma unroll
        for (int u = 0; u < CH; u++)
            if (DEG > 0 || q0 + u < deg) r[u] = ntload(&Rt[(int64_t)refs[q0 + u] * 64]);
#pragma unroll
        for (int u = 0; u < CH; u++)
            if (DEG > 0 || q0 + u < deg) msum += r[u];
    }
    return msum;
}

// ---- min-sum passes, software-pipelined: while a wave works on one (tile, node) item, the state word, the old
// record and the scalar table entries of its NEXT item are already in flight, and the gathers of the current
// item (Q rows / record words, mostly L2 hits) are issued before the first wait.  The passes were latency-bound
// with everything requested on demand (four to five dependent round trips per item).
template <int DEG, int QCAP>
__device__ __forceinline__ void cn_msa_q(const MsaRec &o, double *__restrict__ rec, const double (&q)[QCAP], int k,
                                         int32_t *st) {
    int sx = 0, imin = 0;
    unsigned neg = 0;
    double m1 = __builtin_huge_val(), m2 = __builtin_huge_val();
#pragma unroll
    for (int j = 0; j < DEG; j++) {
        sx ^= __double2hiint(q[j]);                              // dec_word = out_llrs < 0 (:193, :248)
        double m = q[j];                                         // 1.0 * llr (:199)
        if (k > 0) m = msa_edge(o, j, 1) + q[j];                 // data * -1 + 1.0 * (msg_sum + llr) (:244-245)
        const double a = fabs(m);
        const bool c1 = a < m1;
        m2 = min_f64(m2, c1 ? m1 : a);
        m1 = min_f64(m1, a);
        imin = c1 ? j : imin;
        neg |= (m < 0.0) ? (1u << j) : 0u;
    }
    if (sx < 0) *st = k + 1;                                     // odd row: iteration k is executed (:205)
    const int lo = imin | ((__popc(neg) & 1) << 8);
    ntstore(m1, &rec[0]);
    ntstore(m2, &rec[64]);
    ntstore(__hiloint2double((int)neg, lo), &rec[128]);
}

template <int QCAP>
__global__ __launch_bounds__(LB) void ldpc_cn_msa_kernel(Bufs bf, int n_v, int n_c,
                                                         const int32_t *__restrict__ row_ptr,
                                                         const int32_t *__restrict__ row_pad, int cpad, int k) {
    int n_slots, buf;
    effective(bf.ctl, n_slots, buf);
    const int n_tiles = n_slots >> 6;
    double *__restrict__ R = bf.R[buf];
    const double *__restrict__ Q = bf.Q[buf];
    int32_t *__restrict__ state = bf.state[buf];
    const int nG = (n_c + 3) >> 2;
    const int64_t RR = 3 * (int64_t)n_c;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int xcd = blockIdx.x & 7, jstride = gridDim.x >> 3;
    struct Pre { int64_t tile; int c, st, deg, live; MsaRec o; int ev[QCAP]; };
    auto stage_a = [&](int idx, Pre &p) -> bool {
        const int tl = idx / nG;
        p.tile = (int64_t)tl * 8 + xcd;
        if (p.tile >= n_tiles) return false;
        p.c = (idx - tl * nG) * 4 + w;
        p.live = p.c < n_c;
        const int cc = p.live ? p.c : 0;
        p.st = state[p.tile * 64 + lane];
        p.deg = row_ptr[cc + 1] - row_ptr[cc];
#pragma unroll
        for (int j = 0; j < QCAP; j++) p.ev[j] = row_pad[(int64_t)cc * cpad + j];
        p.o = MsaRec{0.0, 0.0, 0u, 0, 0};
        if (k > 0) p.o = msa_load(R + (p.tile * RR + (int64_t)cc * 3) * 64 + lane);
        return true;
    };
    Pre nxt;
    int idx = blockIdx.x >> 3;
    bool more = stage_a(idx, nxt);
    while (more) {
        const Pre cur = nxt;
        const double *__restrict__ Qt = Q + cur.tile * n_v * 64 + lane;
        double q[QCAP];
#pragma unroll
        for (int j = 0; j < QCAP; j++) q[j] = Qt[(int64_t)cur.ev[j] * 64];   // entries past deg: valid rows, unused
        idx += jstride;
        more = stage_a(idx, nxt);
        if (cur.live && cur.st >= k) {                            // else frozen, retired or padding
            double *__restrict__ rec = R + (cur.tile * RR + (int64_t)cur.c * 3) * 64 + lane;
            int32_t *stp = &state[cur.tile * 64 + lane];
            switch (cur.deg) {
#define CPX_CASE(D) case D: if (D <= QCAP) { cn_msa_q<(D <= QCAP ? D : 1), QCAP>(cur.o, rec, q, k, stp); break; }
            CPX_CASE(2) CPX_CASE(3) CPX_CASE(4) CPX_CASE(5) CPX_CASE(6) CPX_CASE(7) CPX_CASE(8) CPX_CASE(9)
            CPX_CASE(10) CPX_CASE(11) CPX_CASE(12)
#undef CPX_CASE
            default:
                cn_msa_row<0>(cur.o, rec, Qt, row_pad + (int64_t)cur.c * cpad, cur.deg, k, stp);
                break;
            }
        }
    }
}

// Min-sum variable pass.  (1) state, column descriptor, llr -- addresses depend on the item only; (2) the meta words
// of up to four edges; (3) the one magnitude each meta word selects.  The pass is bound by L2 gather bandwidth
// (measured: loading both magnitudes and selecting in registers saves a round trip but costs 0.73 vs 0.60 ms).
__device__ __forceinline__ void vn_msa_chunk(const double *__restrict__ Rt, int r0, int r1, int r2, int r3, int n,
                                             double &msum) {
    const int ref[4] = {r0, r1, r2, r3};
    double mn[4], m[4];
#pragma unroll
    for (int u = 0; u < 4; u++)
        if (u < n) m[u] = Rt[(int64_t)(ref[u] >> 5) * 192 + 128];
#pragma unroll
    for (int u = 0; u < 4; u++)
        if (u < n) mn[u] = Rt[(int64_t)(ref[u] >> 5) * 192 + (((ref[u] & 31) == (__double2loint(m[u]) & 0xff)) ? 64 : 0)];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        if (u < n) {
            const int j = ref[u] & 31, lo = __double2loint(m[u]);
            const unsigned ng = (((unsigned)__double2hiint(m[u]) >> j) ^ (unsigned)(lo >> 8)) & 1u;
            msum += __hiloint2double(__double2hiint(mn[u]) | (int)(ng << 31), __double2loint(mn[u]));   // increasing check (:243)
        }
    }
}

__global__ __launch_bounds__(LB) void ldpc_vn_msa_kernel(Bufs bf, int n_v, int n_c,
                                                         const int32_t *__restrict__ col_ptr,
                                                         const int32_t *__restrict__ col_pad, int vpad, int k,
                                                         int32_t *__restrict__ iters) {
    int n_slots, buf;
    effective(bf.ctl, n_slots, buf);
    const int n_tiles = n_slots >> 6;
    const double *__restrict__ R = bf.R[buf];
    const double *__restrict__ L = bf.L[buf];
    double *__restrict__ Q = bf.Q[buf];
    const int32_t *__restrict__ state = bf.state[buf];
    const int nG = (n_v + 3) >> 2;
    const int64_t RR = 3 * (int64_t)n_c;
    CPX_ITEM_LOOP(nG) {
        const int tl = idx / nG, vg = idx - tl * nG;
        const int64_t tile = (int64_t)tl * 8 + xcd;
        if (tile >= n_tiles) break;
        const int v = vg * 4 + w;
        if (v >= n_v) continue;
        const int64_t slot = tile * 64 + lane;
        const int st = state[slot];
        const int deg = col_ptr[v + 1] - col_ptr[v];
        const int32_t *__restrict__ refs = col_pad + (int64_t)v * vpad;
        const int r0 = refs[0], r1 = refs[1], r2 = refs[2], r3 = refs[3];
        const int64_t i = (tile * n_v + v) * 64 + lane;
        const double l = ntload(&L[i]);
        if (st <= k) continue;                                    // syndrome was zero at the start of iteration k
        const double *__restrict__ Rt = R + tile * RR * 64 + lane;
        double msum = 0.0;
        vn_msa_chunk(Rt, r0, r1, r2, r3, deg, msum);
        for (int q0 = 4; q0 < deg; q0 += 4)
            vn_msa_chunk(Rt, refs[q0], refs[q0 + 1], refs[q0 + 2], refs[q0 + 3], deg - q0, msum);
        Q[i] = msum + l;                                          // msg_sum + llr (:245, :247)
        if (v == 0 && iters) iters[bf.orig[buf][slot]] += 1;
    }
}

// Sum-product variable pass of iteration k (:243-248): Q = column sum + llr.
__global__ __launch_bounds__(LB) void ldpc_vn_spa_kernel(Bufs bf, int n_v, int64_t RR,
                                                     const int32_t *__restrict__ col_ptr,
                                                     const int32_t *__restrict__ col_pad, int vpad, int k,
                                                     int32_t *__restrict__ iters) {
    int n_slots, buf;
    effective(bf.ctl, n_slots, buf);
    const int n_tiles = n_slots >> 6;
    const double *__restrict__ R = bf.R[buf];
    const double *__restrict__ L = bf.L[buf];
    double *__restrict__ Q = bf.Q[buf];
    const int32_t *__restrict__ state = bf.state[buf];
    const int nG = (n_v + 3) >> 2;
    CPX_ITEM_LOOP(nG) {
        const int tl = idx / nG, vg = idx - tl * nG;
        const int64_t tile = (int64_t)tl * 8 + xcd;
        if (tile >= n_tiles) break;
        const int v = vg * 4 + w;
        if (v >= n_v) continue;
        const int64_t slot = tile * 64 + lane;
        const int st = state[slot];
        const int deg = col_ptr[v + 1] - col_ptr[v];
        const int32_t *__restrict__ refs = col_pad + (int64_t)v * vpad;
        const int64_t i = (tile * n_v + v) * 64 + lane;
        const double l = ntload(&L[i]);
        if (st <= k) continue;                                    // syndrome was zero at the start of iteration k
        const double *__restrict__ Rt = R + tile * RR * 64 + lane;
        double msum;
        switch (deg) {
        case 2: msum = vn_sum<2>(Rt, refs, deg); break;
        case 3: msum = vn_sum<3>(Rt, refs, deg); break;
        case 4: msum = vn_sum<4>(Rt, refs, deg); break;
        case 5: msum = vn_sum<5>(Rt, refs, deg); break;
        case 6: msum = vn_sum<6>(Rt, refs, deg); break;
        case 7: msum = vn_sum<7>(Rt, refs, deg); break;
        case 8: msum = vn_sum<8>(Rt, refs, deg); break;
        default: msum = vn_sum<0>(Rt, refs, deg); break;
        }
        Q[i] = msum + l;                                          // msg_sum + llr (:245, :247)
        if (v == 0 && iters) iters[bf.orig[buf][slot]] += 1;
    }
}

// After iteration k: commit a finished move, count the slots that go on (state == k+1) and, when they are at
// most half of the live slots, number them densely (dst) and request a move.  One workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void ldpc_scan_kernel(Bufs bf, int k, int patient) {
    __shared__ int part[1024];
    __shared__ int hdr[2];
    Ctl *ctl = bf.ctl;
    if (threadIdx.x == 0) {
        if (ctl->do_move) { ctl->n_slots = ctl->n_new; ctl->buf ^= 1; ctl->do_move = 0; }
        hdr[0] = ctl->n_slots; hdr[1] = ctl->buf;
    }
    __syncthreads();
    const int n = hdr[0], buf = hdr[1];
    if (n <= 64) return;
    const int32_t *__restrict__ state = bf.state[buf];
    const int per = (n + 1023) >> 10;
    const int s0 = threadIdx.x * per, s1 = min(n, s0 + per);
    int cnt = 0;
    for (int s = s0; s < s1; s++) cnt += state[s] > k;
    part[threadIdx.x] = cnt;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {                    // inclusive scan of the per-thread counts
        const int add = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    const int total = part[1023];
    // A move pays when the live set is at most half full.  For min-sum (`patient`) a move costs about three
    // iterations of the slots it moves, so it also waits until the set has stopped shrinking quickly (< 20 % since
    // the last iteration) or is at most a quarter full: while the set is collapsing, a later move is cheaper.
    // For sum-product an iteration costs more than a move and the move is never delayed.
    const int prev = ctl->prev_active;
    __syncthreads();
    if (threadIdx.x == 0) ctl->prev_active = total;
    if (total * 2 > n) return;
    if (patient && total * 4 > n && (int64_t)total * 5 < (int64_t)prev * 4) return;
    int run = part[threadIdx.x] - cnt;
    for (int s = s0; s < s1; s++) bf.dst[s] = state[s] > k ? run++ : -1;
    const int n_new = (total + 63) & ~63;
    if (threadIdx.x < 64 && total + (int)threadIdx.x < n_new) bf.state[buf ^ 1][total + threadIdx.x] = -1;   // padding
    if (threadIdx.x == 0) { ctl->n_new = n_new; ctl->do_move = 1; }
}

// Move pass (only when requested): columns of continuing slots go to their dense position in the other buffer
// set; frozen slots are retired to the caller's out_llrs / dec_word ([n_v][B], reference layout :251-253).
// (sv, sb): element strides of out / dec per variable and per block -- (B, 1) for the [n_v][B] layout, (1, n_v) for block-major
__global__ __launch_bounds__(LB) void ldpc_move_kernel(Bufs bf, int n_v, int64_t E /* rows of R per tile */, int k, int64_t B,
                                                       double *__restrict__ out, int8_t *__restrict__ dec, int64_t sv, int64_t sb) {
    const Ctl *ctl = bf.ctl;
    if (!ctl->do_move) return;
    const int buf = ctl->buf, n_tiles = ctl->n_slots >> 6;
    const int32_t *__restrict__ state = bf.state[buf];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t rows = E + 2 * (int64_t)n_v + 1;
    constexpr int RPW = 16;                                       // rows per wave and item: 16 row loads in flight
    const int64_t nRG = (rows + 4 * RPW - 1) / (4 * RPW);
    const int64_t items = nRG * n_tiles;
    for (int64_t idx = blockIdx.x; idx < items; idx += gridDim.x) {
        const int64_t tile = idx / nRG, row0 = ((idx - tile * nRG) * 4 + w) * RPW;
        if (row0 >= rows) continue;
        const int64_t slot = tile * 64 + lane;
        const int st = state[slot];
        if (st < 0) continue;
        const bool go = st > k;
        const int d = go ? bf.dst[slot] : 0;
        const int64_t dt = d >> 6, dl = d & 63;
        const int64_t o = go ? 0 : bf.orig[buf][slot];
        double x[RPW];
#pragma unroll
        for (int u = 0; u < RPW; u++) {
            const int64_t row = row0 + u;
            if (row < E) x[u] = bf.R[buf][(tile * E + row) * 64 + lane];
            else if (row < E + n_v) x[u] = bf.Q[buf][(tile * n_v + (row - E)) * 64 + lane];
            else if (row < E + 2 * (int64_t)n_v) x[u] = bf.L[buf][(tile * n_v + (row - E - n_v)) * 64 + lane];
        }
#pragma unroll
        for (int u = 0; u < RPW; u++) {
            const int64_t row = row0 + u;
            if (row < E) {
                if (go) bf.R[buf ^ 1][(dt * E + row) * 64 + dl] = x[u];
            } else if (row < E + n_v) {
                const int64_t v = row - E;
                if (go) {
                    bf.Q[buf ^ 1][(dt * n_v + v) * 64 + dl] = x[u];
                } else {
                    out[v * sv + o * sb] = x[u];                                      // (:247)
                    dec[v * sv + o * sb] = (int8_t)(__builtin_signbit(x[u]) ? 1 : 0); // (:248)
                }
            } else if (row < E + 2 * (int64_t)n_v) {
                if (go) bf.L[buf ^ 1][(dt * n_v + (row - E - n_v)) * 64 + dl] = x[u];
            } else if (row < rows && go) {
                bf.state[buf ^ 1][d] = st;
                bf.orig[buf ^ 1][d] = bf.orig[buf][slot];
            }
        }
    }
}

// End of the decode: every slot still in the working set is retired.  A wave takes 16 consecutive variables of its tile: in the
// block-major layout a lane then writes 128 contiguous bytes of its block's row (one variable per wave wrote 8 bytes per 15 KB
// stride: 1.8 GB of traffic for 0.54 GB of data, 0.78 ms for 16 384 blocks of (1944,1296)).
__global__ __launch_bounds__(LB) void ldpc_final_kernel(Bufs bf, int n_v, int64_t B, double *__restrict__ out,
                                                        int8_t *__restrict__ dec, int64_t sv, int64_t sb) {
    int n_slots, buf;
    effective(bf.ctl, n_slots, buf);
    const int n_tiles = n_slots >> 6;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int RPW = 16;
    const int64_t nVG = (n_v + 4 * RPW - 1) / (4 * RPW);
    const int64_t items = nVG * n_tiles;
    for (int64_t idx = blockIdx.x; idx < items; idx += gridDim.x) {
        const int64_t tile = idx / nVG, v0 = ((idx - tile * nVG) * 4 + w) * RPW;
        if (v0 >= n_v) continue;
        const int64_t slot = tile * 64 + lane;
        if (bf.state[buf][slot] < 0) continue;
        const int64_t o = bf.orig[buf][slot];
        double x[RPW];
#pragma unroll
        for (int u = 0; u < RPW; u++)
            if (v0 + u < n_v) x[u] = bf.Q[buf][(tile * n_v + v0 + u) * 64 + lane];
#pragma unroll
        for (int u = 0; u < RPW; u++)
            if (v0 + u < n_v) {
                out[(v0 + u) * sv + o * sb] = x[u];
                dec[(v0 + u) * sv + o * sb] = (int8_t)(__builtin_signbit(x[u]) ? 1 : 0);
            }
    }
}

// ---- min-sum with a NaN among the LLRs: detect and redo ---------------------------------------------------------------
// NumPy's `min` and `sign` PROPAGATE a NaN (ldpc.py:229-238): a NaN message makes every other message of its check NaN, the
// next variable pass (:243-245) spreads that over every neighbour, and within a few iterations the block's whole connected
// component is NaN -- with dec_word = signbit(NaN) deciding the syndrome test (:205) and thereby the iteration count.  The
// fast kernels' minimum (v_min_f64 returns the other operand) confines a NaN to the messages computed from it.  They only
// DETECT a NaN LLR while loading a block (one flag byte per block) and this kernel decodes the flagged blocks again, a
// literal restatement of the reference's loop with NaN-propagating minimum and sign.  No NaN in the batch: every
// workgroup reads its share of the flag bytes and exits.
// NaN signs: every NaN here is a propagated copy of an input NaN (the clip bounds the values, no operation of the loop creates
// one), and neither x86 nor gfx950 changes the sign of a NaN it propagates; with inputs of ONE NaN sign (np.nan) results are
// the reference's bit for bit; with mixed signs, which NaN an operation of two NaNs returns is unspecified on both sides.
// The same literal kernel is also the GENERAL path: a Tanner graph with a check of more than MAXDEG = 32 edges (the reference
// has no limit, ldpc.py:144-254; the tiled and resident kernels keep a row in registers / 5-bit positions) is decoded by it
// alone, min-sum or sum-product, every block (flags == null).  Sum-product rows use the exact-order sequence of ldpc_dev.h
// (tanh as (1 - e) / (1 + e), sequential product, reciprocal, clip, 2 atanh, clip -- :209-227), the one the fast kernels
// fall back to near saturation.
struct ExactParams {
    const double *llr;        // [B][n_v], already clipped in place
    const uint8_t *flags;     // [B], or null: every block
    double *out;              // element (v, b) at v * sv + b * sb: [n_v][B] or block-major [B][n_v]
    int8_t *dec;
    int64_t sv, sb;
    int32_t *iters;           // [B] or null
    double *scratch;          // per workgroup: M[E] (variable -> check), R[E] (check -> variable), tot[n_v]
    const int32_t *edge_var, *row_ptr, *col_ptr, *col_edge;
    int64_t B, E;
    int n_v, n_c, n_iters;
};

template <bool SPA>
__global__ __launch_bounds__(256) void ldpc_exact_kernel(ExactParams p) {
    __shared__ int any;
    __shared__ unsigned char flagged[256];
    const int tid = threadIdx.x;
    double *M = p.scratch + (int64_t)blockIdx.x * (2 * p.E + p.n_v);
    double *R = M + p.E;
    double *tot = R + p.E;
    const int64_t per = (p.B + gridDim.x - 1) / gridDim.x;        // contiguous share of the blocks
    const int64_t lo = (int64_t)blockIdx.x * per, hi = (lo + per < p.B) ? lo + per : p.B;
    for (int64_t base = lo; base < hi; base += 256) {
        const int mine = (base + tid < hi) ? (p.flags ? p.flags[base + tid] : 1) : 0;
        __syncthreads();                                          // the previous round's readers of `flagged` are done
        flagged[tid] = (unsigned char)mine;
        if (!__syncthreads_or(mine)) continue;                    // nothing flagged among these 256 blocks (the usual case): one barrier
        for (int i = 0; i < 256 && base + i < hi; i++) {
            if (!__builtin_amdgcn_readfirstlane((int)flagged[i])) continue;   // workgroup-uniform: a SCALAR branch around the barriers below
            const int64_t b = base + i;
            const double *l = p.llr + b * p.n_v;
            for (int64_t e = tid; e < p.E; e += 256) M[e] = 1.0 * l[p.edge_var[e]];          // H.multiply(llr) (:199)
            for (int v = tid; v < p.n_v; v += 256) tot[v] = l[v];                             // out_llrs = llr (:194)
            __syncthreads();
            int it = 0;
            for (; it < p.n_iters; it++) {
                if (tid == 0) any = 0;
                __syncthreads();
                for (int c = tid; c < p.n_c; c += 256) {          // H . dec_word % 2 (:205), dec_word = signbit(out_llrs)
                    int par = 0;
                    for (int e = p.row_ptr[c]; e < p.row_ptr[c + 1]; e++) par ^= __builtin_signbit(tot[p.edge_var[e]]) ? 1 : 0;
                    if (par) any = 1;
                }
                __syncthreads();
                if (!__builtin_amdgcn_readfirstlane(any)) break;     // uniform (scalar) exit
                for (int c = tid; c < p.n_c; c += 256) {
                    const int b0 = p.row_ptr[c], deg = p.row_ptr[c + 1] - b0;
                    if (SPA) {                                    // (:209-227), exact-order row
                        double prod = 1.0;
                        for (int j = 0; j < deg; j++) {
                            double se, u, w;
                            spa_in(M[b0 + j], se, u, w);
                            prod *= spa_exact_t(se);
                        }
                        for (int j = 0; j < deg; j++) {
                            double se, u, w;
                            spa_in(M[b0 + j], se, u, w);
                            R[b0 + j] = spa_out_exact(spa_exact_t(se), prod);
                        }
                    } else {                                      // (:231-238)
                        for (int j = 0; j < deg; j++) {
                            double sp = 1.0, mn = __builtin_huge_val();
                            for (int q = 0; q < deg; q++) {
                                if (q == j) continue;
                                const double v = M[b0 + q];
                                double sg = (double)((v > 0.0) - (v < 0.0));          // np.sign; sign(NaN) = that NaN
                                if (v != v) sg = v;
                                sp *= sg;                                             // .prod()
                                const double av = fabs(v);
                                if (av < mn || av != av) mn = av;                     // .min() propagates NaN
                            }
                            R[b0 + j] = sp * mn;
                        }
                    }
                }
                __syncthreads();
                for (int v = tid; v < p.n_v; v += 256) {          // (:243-248)
                    double msum = 0.0;
                    for (int q = p.col_ptr[v]; q < p.col_ptr[v + 1]; q++) msum += R[p.col_edge[q]];   // sum(0): increasing check
                    const double t = msum + l[v];
                    for (int q = p.col_ptr[v]; q < p.col_ptr[v + 1]; q++) {
                        const int e = p.col_edge[q];
                        double m = R[e] * -1.0;                   // data *= -1 (:244)
                        m += 1.0 * t;                             // data += H.multiply(msg_sum + llr).data (:245)
                        M[e] = m;
                    }
                    tot[v] = t;
                }
                __syncthreads();
            }
            for (int v = tid; v < p.n_v; v += 256) {
                const double x = tot[v];
                p.out[(int64_t)v * p.sv + b * p.sb] = x;
                p.dec[(int64_t)v * p.sv + b * p.sb] = (int8_t)(__builtin_signbit(x) ? 1 : 0);
            }
            if (tid == 0 && p.iters) p.iters[b] = it;
            __syncthreads();
        }
    }
}

// in-place clip of the LLRs (ldpc.py:186) for the general path (the other paths clip while loading a block); *clipped = 1
// when a value changed (may be null)
__global__ __launch_bounds__(256) void ldpc_clip_kernel(double *llr, int64_t n, int *clipped) {
    bool ch = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double v = llr[i], c = clip_nan(v, -500.0, 500.0);
        if (c != v && v == v) { llr[i] = c; ch = true; }
    }
    if (clipped && __ballot(ch) != 0 && (threadIdx.x & 63) == 0) *clipped = 1;
}

}  // namespace

extern "C" {

// ---- compiled design ("blob"): everything the device needs about a Tanner graph, position-independent ------------------
// Layout (little endian, int32 words after the header):
//   header   magic "CPXLDPC1", version, n_v, n_c, max_cdeg, max_vdeg, cpad, vpad, n_edges, payload words, FNV-1a of the payload
//   payload  edge_check[E] edge_var[E] row_ptr[n_c+1] col_ptr[n_v+1] col_edge[E] col_cj[E]
//            row_pad[n_c*cpad+16] col_pad_edge[n_v*vpad+16] col_pad_cj[n_v*vpad+16]
// Built on the host without a device (cpx_ldpc_blob_build), so a design file can be compiled once and the blob cached
// next to it, keyed by the file's hash (commpy_amd/channelcoding/ldpc.py); cpx_ldpc_create_from_blob validates and uploads.
struct LdpcBlobHeader {
    char magic[8];
    uint32_t version;
    int32_t n_v, n_c, max_cdeg, max_vdeg, cpad, vpad;
    int32_t reserved;
    int64_t n_edges;
    uint64_t payload_words;
    uint64_t checksum;
};
static_assert(sizeof(LdpcBlobHeader) == 64, "blob header layout");
static const char LDPC_BLOB_MAGIC[8] = {'C', 'P', 'X', 'L', 'D', 'P', 'C', '1'};

static uint64_t fnv1a(const void *data, size_t n) {
    const unsigned char *p = static_cast<const unsigned char *>(data);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

struct LdpcBlobView {                       // pointers into a validated blob
    const LdpcBlobHeader *h;
    const int32_t *edge_check, *edge_var, *row_ptr, *col_ptr, *col_edge, *col_cj, *row_pad, *col_pad_edge, *col_pad_cj;
    size_t n_row_pad, n_col_pad;
};

static size_t blob_payload_words(int64_t E, int n_v, int n_c, int cpad, int vpad) {
    return (size_t)(4 * E + (n_c + 1) + (n_v + 1)) + ((size_t)n_c * cpad + 16) + 2 * ((size_t)n_v * vpad + 16);
}

static int blob_view(const void *blob, size_t nbytes, LdpcBlobView &v) {
    CPX_REQUIRE(blob && nbytes >= sizeof(LdpcBlobHeader), CPX_EINVAL, "ldpc blob: too short");
    CPX_REQUIRE(((uintptr_t)blob & 3) == 0, CPX_EINVAL, "ldpc blob: buffer must be 4-byte aligned");
    const LdpcBlobHeader *h = static_cast<const LdpcBlobHeader *>(blob);
    CPX_REQUIRE(memcmp(h->magic, LDPC_BLOB_MAGIC, 8) == 0 && h->version == 1, CPX_EINVAL, "ldpc blob: bad magic / version");
    CPX_REQUIRE(h->n_v > 0 && h->n_c > 0 && h->n_edges > 0 && h->n_edges < (1ll << 30) && h->n_v < (1 << 24) && h->n_c < (1 << 24),
                CPX_EINVAL, "ldpc blob: bad dimensions");
    CPX_REQUIRE(h->max_cdeg >= 1 && h->max_vdeg >= 1 && h->cpad == ((h->max_cdeg + 3) & ~3) &&
                h->vpad == ((h->max_vdeg + 3) & ~3), CPX_EINVAL, "ldpc blob: bad degrees");
    const size_t words = blob_payload_words(h->n_edges, h->n_v, h->n_c, h->cpad, h->vpad);
    CPX_REQUIRE(h->payload_words == words && nbytes == sizeof(LdpcBlobHeader) + 4 * words, CPX_EINVAL, "ldpc blob: size mismatch");
    const int32_t *w = reinterpret_cast<const int32_t *>(h + 1);
    CPX_REQUIRE(fnv1a(w, 4 * words) == h->checksum, CPX_EINVAL, "ldpc blob: checksum mismatch (corrupted cache file?)");
    const int64_t E = h->n_edges;
    v.h = h;
    v.edge_check = w; w += E;
    v.edge_var = w; w += E;
    v.row_ptr = w; w += h->n_c + 1;
    v.col_ptr = w; w += h->n_v + 1;
    v.col_edge = w; w += E;
    v.col_cj = w; w += E;
    v.n_row_pad = (size_t)h->n_c * h->cpad + 16;
    v.n_col_pad = (size_t)h->n_v * h->vpad + 16;
    v.row_pad = w; w += v.n_row_pad;
    v.col_pad_edge = w; w += v.n_col_pad;
    v.col_pad_cj = w;
    // the kernels index device memory with these tables: range-check all of them
    CPX_REQUIRE(v.row_ptr[0] == 0 && v.row_ptr[h->n_c] == E && v.col_ptr[0] == 0 && v.col_ptr[h->n_v] == E, CPX_EINVAL,
                "ldpc blob: bad row/column pointers");
    for (int c = 0; c < h->n_c; c++) {
        const int d = v.row_ptr[c + 1] - v.row_ptr[c];
        CPX_REQUIRE(d >= 0 && d <= h->max_cdeg, CPX_EINVAL, "ldpc blob: bad check degree");
    }
    for (int q = 0; q < h->n_v; q++) {
        const int d = v.col_ptr[q + 1] - v.col_ptr[q];
        CPX_REQUIRE(d >= 0 && d <= h->max_vdeg, CPX_EINVAL, "ldpc blob: bad variable degree");
    }
    for (int64_t e = 0; e < E; e++) {
        CPX_REQUIRE(v.edge_check[e] >= 0 && v.edge_check[e] < h->n_c && v.edge_var[e] >= 0 && v.edge_var[e] < h->n_v &&
                    v.col_edge[e] >= 0 && v.col_edge[e] < E && (v.col_cj[e] >> 5) >= 0 && (v.col_cj[e] >> 5) < h->n_c &&
                    (v.col_cj[e] & 31) < std::max(h->max_cdeg, 1), CPX_EINVAL, "ldpc blob: index out of range");
    }
    for (size_t i = 0; i < v.n_row_pad; i++)
        CPX_REQUIRE(v.row_pad[i] >= 0 && v.row_pad[i] < h->n_v, CPX_EINVAL, "ldpc blob: padded row entry out of range");
    for (size_t i = 0; i < v.n_col_pad; i++)
        CPX_REQUIRE(v.col_pad_edge[i] >= 0 && v.col_pad_edge[i] < E && (v.col_pad_cj[i] >> 5) >= 0 &&
                    (v.col_pad_cj[i] >> 5) < h->n_c, CPX_EINVAL, "ldpc blob: padded column entry out of range");
    // (check, position) pairs become LDS addresses of the resident path (roff + 8 (check * rstride + position)): a position
    // must lie inside ITS check's row, and the views must describe the same graph as the edge list (FNV is no protection
    // against a deliberately edited cache file)
    for (int64_t e = 0; e < E; e++) {
        const int c = v.edge_check[e];
        CPX_REQUIRE(e >= v.row_ptr[c] && e < v.row_ptr[c + 1], CPX_EINVAL, "ldpc blob: edge list and row pointers disagree");
    }
    for (int c = 0; c < h->n_c; c++)
        for (int j = 0; j < v.row_ptr[c + 1] - v.row_ptr[c]; j++)
            CPX_REQUIRE(v.row_pad[(size_t)c * h->cpad + j] == v.edge_var[v.row_ptr[c] + j], CPX_EINVAL,
                        "ldpc blob: padded rows and edge list disagree");
    for (int q = 0; q < h->n_v; q++)
        for (int i = v.col_ptr[q]; i < v.col_ptr[q + 1]; i++) {
            const int e = v.col_edge[i], cj = v.col_cj[i], c = cj >> 5, pos = cj & 31;
            CPX_REQUIRE(v.edge_var[e] == q && v.edge_check[e] == c && pos == ((e - v.row_ptr[c]) & 31), CPX_EINVAL,
                        "ldpc blob: column view and edge list disagree");
            const size_t k = (size_t)q * h->vpad + (size_t)(i - v.col_ptr[q]);
            CPX_REQUIRE(v.col_pad_edge[k] == e && v.col_pad_cj[k] == cj, CPX_EINVAL, "ldpc blob: padded columns and column view disagree");
        }
    return CPX_OK;
}

int cpx_ldpc_blob_build(int n_vnodes, int n_cnodes, int64_t n_edges, const int32_t *edge_check, const int32_t *edge_var,
                        void *blob, size_t cap, size_t *need) {
    CPX_REQUIRE(edge_check && edge_var && need, CPX_EINVAL, "cpx_ldpc_blob_build: null pointer");
    CPX_REQUIRE(n_vnodes > 0 && n_cnodes > 0 && n_edges > 0, CPX_EINVAL, "cpx_ldpc_create: empty code");
    CPX_REQUIRE(n_edges < (1ll << 30), CPX_ELIMIT, "cpx_ldpc_create: too many edges");
    CPX_REQUIRE(n_vnodes < (1 << 24) && n_cnodes < (1 << 24), CPX_ELIMIT, "cpx_ldpc_create: code too large (n < 2^24)");
    const int64_t E = n_edges;
    std::vector<int32_t> row_ptr(n_cnodes + 1, 0), col_ptr(n_vnodes + 1, 0);
    for (int64_t e = 0; e < E; e++) {
        CPX_REQUIRE(edge_check[e] >= 0 && edge_check[e] < n_cnodes && edge_var[e] >= 0 && edge_var[e] < n_vnodes,
                    CPX_EINVAL, "cpx_ldpc_create: edge %lld out of range", (long long)e);
        if (e > 0) {
            const bool sorted = edge_check[e] > edge_check[e - 1] ||
                                (edge_check[e] == edge_check[e - 1] && edge_var[e] > edge_var[e - 1]);
            CPX_REQUIRE(sorted, CPX_EINVAL, "cpx_ldpc_create: edges must be strictly sorted by (check, variable)");
        }
        row_ptr[edge_check[e] + 1]++;
        col_ptr[edge_var[e] + 1]++;
    }
    int max_cdeg = 0, max_vdeg = 0;
    for (int c = 0; c < n_cnodes; c++) { max_cdeg = std::max(max_cdeg, row_ptr[c + 1]); row_ptr[c + 1] += row_ptr[c]; }
    for (int v = 0; v < n_vnodes; v++) { max_vdeg = std::max(max_vdeg, col_ptr[v + 1]); col_ptr[v + 1] += col_ptr[v]; }
    // a check of more than MAXDEG edges: the 5-bit position field below wraps -- such a code is only ever decoded by
    // ldpc_exact_kernel, which does not read it
    const int cpad = (max_cdeg + 3) & ~3, vpad = (max_vdeg + 3) & ~3;
    const size_t words = blob_payload_words(E, n_vnodes, n_cnodes, cpad, vpad);
    *need = sizeof(LdpcBlobHeader) + 4 * words;
    if (!blob) return CPX_OK;                                                    // size query
    CPX_REQUIRE(cap >= *need, CPX_EINVAL, "cpx_ldpc_blob_build: buffer too small (%zu < %zu)", cap, *need);
    CPX_REQUIRE(((uintptr_t)blob & 7) == 0, CPX_EINVAL, "cpx_ldpc_blob_build: buffer must be 8-byte aligned");
    LdpcBlobHeader *h = static_cast<LdpcBlobHeader *>(blob);
    memset(h, 0, sizeof(*h));
    memcpy(h->magic, LDPC_BLOB_MAGIC, 8);
    h->version = 1; h->n_v = n_vnodes; h->n_c = n_cnodes; h->max_cdeg = max_cdeg; h->max_vdeg = max_vdeg;
    h->cpad = cpad; h->vpad = vpad; h->n_edges = E; h->payload_words = words;
    int32_t *w = reinterpret_cast<int32_t *>(h + 1);
    memset(w, 0, 4 * words);
    int32_t *b_ec = w; w += E;
    int32_t *b_ev = w; w += E;
    int32_t *b_rp = w; w += n_cnodes + 1;
    int32_t *b_cp = w; w += n_vnodes + 1;
    int32_t *b_ce = w; w += E;
    int32_t *b_cj = w; w += E;
    int32_t *b_rpad = w; w += (size_t)n_cnodes * cpad + 16;       // + 16: the pipelined passes read a fixed number of
    int32_t *b_cpe = w; w += (size_t)n_vnodes * vpad + 16;         // entries per row, past the last row's end
    int32_t *b_cpc = w;
    memcpy(b_ec, edge_check, 4 * (size_t)E);
    memcpy(b_ev, edge_var, 4 * (size_t)E);
    memcpy(b_rp, row_ptr.data(), 4 * (size_t)(n_cnodes + 1));
    memcpy(b_cp, col_ptr.data(), 4 * (size_t)(n_vnodes + 1));
    std::vector<int32_t> fill(col_ptr.begin(), col_ptr.end() - 1);
    for (int64_t e = 0; e < E; e++) {                                            // increasing e == increasing check
        const int32_t q = fill[edge_var[e]]++;
        b_ce[q] = (int32_t)e;
        // 5-bit position field: meaningful for checks of up to 32 edges only -- a code beyond that is decoded from edge_var / row_ptr /
        // col_edge by ldpc_exact_kernel alone (cpx_ldpc_bp_decode_* takes that branch before any consumer of col_cj; forced paths: EINVAL)
        b_cj[q] = (edge_check[e] << 5) | ((int32_t)(e - row_ptr[edge_check[e]]) & 31);
    }
    for (int k = 0; k < n_cnodes; k++)
        for (int j = 0; j < row_ptr[k + 1] - row_ptr[k]; j++) b_rpad[(size_t)k * cpad + j] = edge_var[row_ptr[k] + j];
    for (int v = 0; v < n_vnodes; v++)
        for (int q = 0; q < col_ptr[v + 1] - col_ptr[v]; q++) {
            b_cpe[(size_t)v * vpad + q] = b_ce[col_ptr[v] + q];
            b_cpc[(size_t)v * vpad + q] = b_cj[col_ptr[v] + q];
        }
    h->checksum = fnv1a(h + 1, 4 * words);
    return CPX_OK;
}

int cpx_ldpc_blob_info(const void *blob, size_t nbytes, int *n_vnodes, int *n_cnodes, int64_t *n_edges, int *max_cnode_deg,
                       int *max_vnode_deg) {
    LdpcBlobView v;
    int rc = blob_view(blob, nbytes, v);
    if (rc) return rc;
    if (n_vnodes) *n_vnodes = v.h->n_v;
    if (n_cnodes) *n_cnodes = v.h->n_c;
    if (n_edges) *n_edges = v.h->n_edges;
    if (max_cnode_deg) *max_cnode_deg = v.h->max_cdeg;
    if (max_vnode_deg) *max_vnode_deg = v.h->max_vdeg;
    return CPX_OK;
}

int cpx_ldpc_create_from_blob(const void *blob, size_t nbytes, cpx_ldpc **out) {
    CPX_REQUIRE(out, CPX_EINVAL, "cpx_ldpc_create_from_blob: null pointer");
    LdpcBlobView v;
    int rc = blob_view(blob, nbytes, v);
    if (rc) return rc;
    if ((rc = ensure_device())) return rc;
    const LdpcBlobHeader *h = v.h;
    const int64_t E = h->n_edges;
    cpx_ldpc *c = new cpx_ldpc;
    c->n_v = h->n_v; c->n_c = h->n_c; c->n_edges = E; c->max_cdeg = h->max_cdeg; c->max_vdeg = h->max_vdeg;
    c->cpad = h->cpad; c->vpad = h->vpad;
    (void)hipGetDevice(&c->device);
    struct Up { int32_t **dst; const int32_t *src; size_t n; } ups[] = {
        {&c->d_edge_var, v.edge_var, (size_t)E}, {&c->d_row_ptr, v.row_ptr, (size_t)h->n_c + 1},
        {&c->d_col_ptr, v.col_ptr, (size_t)h->n_v + 1}, {&c->d_col_edge, v.col_edge, (size_t)E},
        {&c->d_col_cj, v.col_cj, (size_t)E}, {&c->d_row_pad, v.row_pad, v.n_row_pad},
        {&c->d_col_pad_edge, v.col_pad_edge, v.n_col_pad}, {&c->d_col_pad_cj, v.col_pad_cj, v.n_col_pad}};
    for (auto &u : ups) {
        hipError_t e1 = hipMalloc((void **)u.dst, sizeof(int32_t) * u.n);
        hipError_t e2 = e1 == hipSuccess ? hipMemcpy(*u.dst, u.src, sizeof(int32_t) * u.n, hipMemcpyHostToDevice) : e1;
        if (e2 != hipSuccess) {
            set_error("cpx_ldpc_create: device upload failed: %s", hipGetErrorString(e2));
            cpx_ldpc_destroy(c);
            return CPX_EHIP;
        }
    }
    if ((rc = ldpc_resident_tables(c, v.row_ptr, v.row_pad, v.col_ptr, v.col_pad_cj))) {
        cpx_ldpc_destroy(c);
        return rc;
    }
    *out = c;
    return CPX_OK;
}

int cpx_ldpc_create(int n_vnodes, int n_cnodes, int64_t n_edges, const int32_t *edge_check, const int32_t *edge_var,
                    cpx_ldpc **out) {
    CPX_REQUIRE(out && edge_check && edge_var, CPX_EINVAL, "cpx_ldpc_create: null pointer");
    size_t need = 0;
    int rc = cpx_ldpc_blob_build(n_vnodes, n_cnodes, n_edges, edge_check, edge_var, nullptr, 0, &need);
    if (rc) return rc;
    if ((rc = ensure_device())) return rc;
    std::vector<uint64_t> buf((need + 7) / 8);
    if ((rc = cpx_ldpc_blob_build(n_vnodes, n_cnodes, n_edges, edge_check, edge_var, buf.data(), buf.size() * 8, &need))) return rc;
    return cpx_ldpc_create_from_blob(buf.data(), need, out);
}

int cpx_ldpc_destroy(cpx_ldpc *c) {
    if (!c) return CPX_OK;
    (void)hipFree(c->d_edge_var); (void)hipFree(c->d_row_ptr); (void)hipFree(c->d_col_ptr); (void)hipFree(c->d_col_edge); (void)hipFree(c->d_col_cj);
    (void)hipFree(c->d_row_pad); (void)hipFree(c->d_col_pad_edge); (void)hipFree(c->d_col_pad_cj);
    ldpc_resident_free(c);
    delete c;
    return CPX_OK;
}

// d_clipped (may be null): set to 1 when the in-place clip of (:186) changed any value of d_llr
// block_major: d_dec / d_out are [B][n_v] instead of [n_v][B]
static int ldpc_decode_impl(const cpx_ldpc *c, double *d_llr, int64_t B, int alg, int n_iters, int8_t *d_dec, double *d_out,
                            int block_major, int32_t *d_iters, int *d_clipped, void *stream);

int cpx_ldpc_bp_decode_batch_dev(const cpx_ldpc *c, double *d_llr, int64_t B, int alg, int n_iters, int8_t *d_dec,
                                 double *d_out, int32_t *d_iters, void *stream) {
    CPX_TRACE("cpx_ldpc_bp_decode_batch_dev");
    return ldpc_decode_impl(c, d_llr, B, alg, n_iters, d_dec, d_out, 0, d_iters, nullptr, stream);
}

int cpx_ldpc_bp_decode_batch_bm_dev(const cpx_ldpc *c, double *d_llr, int64_t B, int alg, int n_iters, int8_t *d_dec,
                                    double *d_out, int32_t *d_iters, void *stream) {
    CPX_TRACE("cpx_ldpc_bp_decode_batch_bm_dev");
    return ldpc_decode_impl(c, d_llr, B, alg, n_iters, d_dec, d_out, 1, d_iters, nullptr, stream);
}

static int ldpc_decode_impl(const cpx_ldpc *c, double *d_llr, int64_t B, int alg, int n_iters, int8_t *d_dec, double *d_out,
                            int block_major, int32_t *d_iters, int *d_clipped, void *stream) {
    cpx::IssueGuard issue_guard;
    CPX_REQUIRE(c, CPX_EINVAL, "ldpc: null code");
    if (int rcd = check_handle_device(c->device, "ldpc")) return rcd;
    CPX_REQUIRE(alg == CPX_LDPC_SPA || alg == CPX_LDPC_MSA, CPX_EINVAL,
                "Please input a valid decoder_algorithm string (meanning \"SPA\" or \"MSA\").");
    CPX_REQUIRE(B >= 0 && n_iters >= 0, CPX_EINVAL, "ldpc: negative size");
    if (B == 0) return CPX_OK;
    CPX_REQUIRE(d_llr && d_dec && d_out, CPX_EINVAL, "ldpc: null device pointer");
    CPX_REQUIRE(B <= (1ll << 30), CPX_ELIMIT, "ldpc: batch too large");
    hipStream_t st = pick_stream(stream);
    const int64_t sv = block_major ? 1 : B, sb = block_major ? (int64_t)c->n_v : 1;    // strides of d_out / d_dec (variable, block)
    // min-sum: one flag byte per block, set by the kernels that load the LLRs when they meet a NaN; flagged blocks are
    // decoded again by ldpc_exact_kernel<false> (scratch-arena slots 4 / 5)
    uint8_t *nanflags = nullptr;
    // literal kernel: `flags` selects the blocks (null = all), `wgs` workgroups, each with (2 E + n_v) doubles of scratch
    auto exact = [&](const uint8_t *flags, unsigned wgs, int algo) -> int {
        ExactParams q;
        void *sc = nullptr;
        if (int rcs = workspace(st, 5, sizeof(double) * (size_t)wgs * (size_t)(2 * c->n_edges + c->n_v), &sc)) return rcs;
        q.llr = d_llr; q.flags = flags; q.out = d_out; q.dec = d_dec; q.iters = d_iters; q.scratch = static_cast<double *>(sc);
        q.edge_var = c->d_edge_var; q.row_ptr = c->d_row_ptr; q.col_ptr = c->d_col_ptr; q.col_edge = c->d_col_edge;
        q.B = B; q.E = c->n_edges; q.n_v = c->n_v; q.n_c = c->n_c; q.n_iters = n_iters; q.sv = sv; q.sb = sb;
        if (algo == CPX_LDPC_SPA) hipLaunchKernelGGL(ldpc_exact_kernel<true>, dim3(wgs), dim3(256), 0, st, q);
        else hipLaunchKernelGGL(ldpc_exact_kernel<false>, dim3(wgs), dim3(256), 0, st, q);
        CPX_HIP(hipGetLastError());
        return CPX_OK;
    };
    if (c->max_cdeg > MAXDEG) {                                   // the general path: every block through the literal kernel
        // a FORCED path (cpx_ldpc_set_path: tests and benchmarks that name a kernel) is never substituted silently
        CPX_REQUIRE(ldpc_forced_path() == 0, CPX_EINVAL, "ldpc: path forced (cpx_ldpc_set_path / CPX_LDPC_PATH) but a check of %d > %d edges "
                    "is only served by the literal kernel", c->max_cdeg, MAXDEG);
        if (d_clipped) CPX_HIP(hipMemsetAsync(d_clipped, 0, sizeof(int), st));
        const int64_t n = B * (int64_t)c->n_v;
        hipLaunchKernelGGL(ldpc_clip_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, st, d_llr, n, d_clipped);
        CPX_HIP(hipGetLastError());
        if (int rce = exact(nullptr, (unsigned)std::min<int64_t>(B, 4 * device_cus()), alg)) return rce;
        note_kernel("ldpc_exact_kernel<%s> (check degree %d > %d)", alg == CPX_LDPC_SPA ? "SPA" : "MSA", c->max_cdeg, MAXDEG);
        return CPX_OK;
    }
    // the redo launch behind a min-sum decode: with no NaN in the batch its workgroups read their share of the flag bytes and
    // exit, so it gets a SMALL grid (64 workgroups: 8 MB of scratch for the (1944,1296) code, was 2 per CU = 37 MB)
    auto msa_redo = [&]() -> int {
        if (!nanflags) return CPX_OK;
        return exact(nanflags, (unsigned)std::min<int64_t>(B, 64), CPX_LDPC_MSA);
    };
    if (alg == CPX_LDPC_MSA) {
        void *w = nullptr;
        if (int rcf = workspace(st, 4, (size_t)B, &w)) return rcf;
        nanflags = static_cast<uint8_t *>(w);
    }
    {   // the whole decoder state of a block in LDS, one persistent launch (ldpc_resident.hip) -- unless it does not fit
        int rcr = CPX_OK;
        if (ldpc_resident_path(c, d_llr, B, alg, n_iters, d_dec, d_out, block_major, d_iters, d_clipped, nanflags, st, &rcr)) {
            if (rcr == CPX_OK) {
                char name[160];
                snprintf(name, sizeof(name), "%s", last_kernel_name());
                rcr = msa_redo();
                note_kernel("%s", name);
            }
            return rcr;
        }
    }
    if (nanflags) CPX_HIP(hipMemsetAsync(nanflags, 0, (size_t)B, st));
    const int64_t E = c->n_edges, nv = c->n_v;
    const int64_t n_tiles = (B + 63) / 64, S = n_tiles * 64;
    const int64_t RR = alg == CPX_LDPC_MSA ? 3 * (int64_t)c->n_c : E;      // rows of R per tile (records / edges)
    CPX_REQUIRE((n_tiles / 8 + 1) * ((std::max<int64_t>(nv, c->n_c) + 3) / 4) < (1ll << 31), CPX_ELIMIT,
                "ldpc: batch x code too large for one launch");
    // one slab: 2 x (R, Q, L) + slot tables + control block
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t szR = al(sizeof(double) * (size_t)(RR * S)), szV = al(sizeof(double) * (size_t)(nv * S)),
                 szI = al(sizeof(int32_t) * (size_t)S);
    char *slab = nullptr;
    int rcw;
    if ((rcw = workspace(st, 0, 2 * szR + 4 * szV + 5 * szI + 256, (void **)&slab))) return rcw;
    Bufs bf;
    char *p = slab;
    for (int i = 0; i < 2; i++) { bf.R[i] = (double *)p; p += szR; }
    for (int i = 0; i < 2; i++) { bf.Q[i] = (double *)p; p += szV; }
    for (int i = 0; i < 2; i++) { bf.L[i] = (double *)p; p += szV; }
    for (int i = 0; i < 2; i++) { bf.state[i] = (int32_t *)p; p += szI; }
    for (int i = 0; i < 2; i++) { bf.orig[i] = (int32_t *)p; p += szI; }
    bf.dst = (int32_t *)p; p += szI;
    bf.ctl = (Ctl *)p;
    hipLaunchKernelGGL(ldpc_init_kernel, dim3((unsigned)n_tiles, (unsigned)((nv + 63) / 64)), dim3(LB), 0, st, d_llr, B,
                       (int)nv, bf, d_iters, d_clipped, nanflags);
    // persistent grids: what is resident at once (occupancy x CUs), no more workgroups than wave items, a
    // multiple of 8 (one share per XCD)
    auto pgrid = [&](const void *fn, int64_t items) {
        const int64_t resident = resident_blocks(fn, LB);
        return (unsigned)std::max<int64_t>(8, std::min<int64_t>(resident, items) / 8 * 8);
    };
    const int qcap = std::min(12, c->cpad);
    const void *f_cn = alg == CPX_LDPC_SPA ? (const void *)ldpc_cn_spa_kernel
                       : qcap == 4        ? (const void *)ldpc_cn_msa_kernel<4>
                       : qcap == 8        ? (const void *)ldpc_cn_msa_kernel<8>
                                          : (const void *)ldpc_cn_msa_kernel<12>;
    const void *f_vn = alg == CPX_LDPC_SPA ? (const void *)ldpc_vn_spa_kernel
                                          : (const void *)ldpc_vn_msa_kernel;
    const unsigned g_cn = pgrid(f_cn, n_tiles * ((c->n_c + 3) / 4)), g_vn = pgrid(f_vn, n_tiles * ((nv + 3) / 4));
    const unsigned g_mv = pgrid((const void *)ldpc_move_kernel, n_tiles * ((RR + 2 * nv + 64) / 64));
    for (int it = 0; it < n_iters; it++) {
        if (alg == CPX_LDPC_SPA) {
            hipLaunchKernelGGL(ldpc_cn_spa_kernel, dim3(g_cn), dim3(LB), 0, st, bf, (int)nv, c->n_c, RR,
                               c->d_row_ptr, c->d_row_pad, c->cpad, it, ldpc_spa_exact() ? 1 : 0);
            hipLaunchKernelGGL(ldpc_vn_spa_kernel, dim3(g_vn), dim3(LB), 0, st, bf, (int)nv, RR, c->d_col_ptr,
                               c->d_col_pad_edge, c->vpad, it, d_iters);
        } else {
#define CN_MSA(QC) hipLaunchKernelGGL((ldpc_cn_msa_kernel<QC>), dim3(g_cn), dim3(LB), 0, st, bf, (int)nv, c->n_c, c->d_row_ptr, c->d_row_pad, c->cpad, it)
#define VN_MSA() hipLaunchKernelGGL(ldpc_vn_msa_kernel, dim3(g_vn), dim3(LB), 0, st, bf, (int)nv, c->n_c, c->d_col_ptr, c->d_col_pad_cj, c->vpad, it, d_iters)
            if (qcap == 4) CN_MSA(4); else if (qcap == 8) CN_MSA(8); else CN_MSA(12);
            VN_MSA();
#undef CN_MSA
#undef VN_MSA
        }
        if (S > 64 && it + 1 < n_iters) {
            hipLaunchKernelGGL(ldpc_scan_kernel, dim3(1), dim3(1024), 0, st, bf, it, alg == CPX_LDPC_MSA ? 1 : 0);
            hipLaunchKernelGGL(ldpc_move_kernel, dim3(g_mv), dim3(LB), 0, st, bf, (int)nv, RR, it, B, d_out, d_dec, sv, sb);
        }
    }
    hipLaunchKernelGGL(ldpc_final_kernel, dim3(g_vn), dim3(LB), 0, st, bf, (int)nv, B, d_out, d_dec, sv, sb);
    CPX_HIP(hipGetLastError());
    if (int rcx = msa_redo()) return rcx;
    note_kernel("ldpc_cn_%s_kernel + ldpc_vn_%s_kernel (tiled)", alg == CPX_LDPC_MSA ? "msa" : "spa", alg == CPX_LDPC_MSA ? "msa" : "spa");
    return CPX_OK;
}

static int ldpc_decode_host(const cpx_ldpc *c, double *llr, int64_t B, int alg, int n_iters, int8_t *dec_word,
                            double *out_llrs, int32_t *iters_done, int block_major) {
    CPX_REQUIRE(c && (llr || B == 0), CPX_EINVAL, "ldpc: null pointer");
    int rc = ensure_device();
    if (rc) return rc;
    if (B == 0) return CPX_OK;
    const size_t nvb = (size_t)((int64_t)c->n_v * B);
    DevBuf d_llr, d_dec, d_out, d_it, d_flag;
    if ((rc = d_llr.alloc(sizeof(double) * nvb))) return rc;
    if ((rc = d_dec.alloc(nvb))) return rc;
    if ((rc = d_out.alloc(sizeof(double) * nvb))) return rc;
    if ((rc = d_it.alloc(sizeof(int32_t) * (size_t)B))) return rc;
    if ((rc = d_flag.alloc(sizeof(int)))) return rc;
    hipStream_t st = lib_stream();
    CPX_HIP(hipMemsetAsync(d_flag.p, 0, sizeof(int), st));
    {
        CPX_TRACE("H2D llr");
        CPX_HIP(hipMemcpyAsync(d_llr.p, llr, sizeof(double) * nvb, hipMemcpyHostToDevice, st));
        if (trace_enabled()) CPX_HIP(hipStreamSynchronize(st));
    }
    rc = ldpc_decode_impl(c, d_llr.as<double>(), B, alg, n_iters, d_dec.as<int8_t>(), d_out.as<double>(), block_major,
                          d_it.as<int32_t>(), d_flag.as<int>(), st);
    if (rc) return rc;
    // in-place clip (:186): the caller's array only changes if a value lay outside +-500 (or is NaN) -- the kernels say so,
    // and the 8-byte-per-LLR copy back (a third of this call's PCIe traffic) is skipped otherwise
    int clipped = 0;
    {
        CPX_TRACE("decode (wait) + clip flag");
        CPX_HIP(hipMemcpyAsync(&clipped, d_flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
        CPX_HIP(hipStreamSynchronize(st));
    }
    CPX_TRACE("D2H results");
    if (clipped && (rc = d2h_pageable(llr, d_llr.p, sizeof(double) * nvb, st))) return rc;
    if (dec_word && (rc = d2h_pageable(dec_word, d_dec.p, nvb, st))) return rc;
    if (out_llrs && (rc = d2h_pageable(out_llrs, d_out.p, sizeof(double) * nvb, st))) return rc;
    if (iters_done) CPX_HIP(hipMemcpyAsync(iters_done, d_it.p, sizeof(int32_t) * (size_t)B, hipMemcpyDeviceToHost, st));
    CPX_HIP(hipStreamSynchronize(st));
    return CPX_OK;
}

int cpx_ldpc_bp_decode_batch(const cpx_ldpc *c, double *llr, int64_t B, int alg, int n_iters, int8_t *dec_word,
                             double *out_llrs, int32_t *iters_done) {
    CPX_TRACE("cpx_ldpc_bp_decode_batch");
    return ldpc_decode_host(c, llr, B, alg, n_iters, dec_word, out_llrs, iters_done, 0);
}

int cpx_ldpc_bp_decode_batch_bm(const cpx_ldpc *c, double *llr, int64_t B, int alg, int n_iters, int8_t *dec_word,
                                double *out_llrs, int32_t *iters_done) {
    CPX_TRACE("cpx_ldpc_bp_decode_batch_bm");
    return ldpc_decode_host(c, llr, B, alg, n_iters, dec_word, out_llrs, iters_done, 1);
}

}  // extern "C"
