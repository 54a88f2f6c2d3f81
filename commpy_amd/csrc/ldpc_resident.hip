// LDPC belief propagation with the whole decoder state of a block resident in LDS (gfx950: 160 KB per CU).
// Same contract and same per-edge float64 operations as ldpc.hip, which documents the formulation
//   ldpc_bp_decode  (/root/reference/commpy/channelcoding/ldpc.py:144-254)
// and stays the path for codes whose state does not fit (and for n_iters == 0).
//
// Why.  The tiled path keeps R/Q/L of 64 blocks per wavefront in HBM and runs one launch per pass: an iteration gathers
// 2 E rows per tile through L2 (latency-bound: 0.4 + 0.6 ms per iteration at B = 32 768), frozen blocks hold lanes until a
// move pass compacts the working set, and a decode is 4 launches x n_iters.  Here ONE persistent launch does everything:
//   * a workgroup owns G = 2^LG block slots; per slot the a-posteriori LLRs Q[n_v] and the check->variable state
//     (min-sum: a 24-byte record per check; sum-product: one float64 per edge) live in LDS for the block's whole life.
//     (1944,1296): 31 KB per block for min-sum -> G = 4; 71 KB for sum-product -> G = 2.
//   * thread = (node, slot).  The check pass walks the checks, the variable pass the variables; both gather from LDS
//     (~100 cycles, hidden by the 14-16 waves of the workgroup) instead of L2.  Only the channel LLR of the variable
//     pass (`+ llr`, :245) is re-read from HBM/L2: 8 B per variable and iteration.
//   * every slot has its own iteration counter.  After a check pass a slot whose syndrome is zero (:203-206) -- or that
//     has used n_iters iterations -- is retired to a block-major staging buffer and REFILLED with the next block of a
//     global queue: continuous batching replaces the scan / move / compaction kernels, and nobody waits for the slowest
//     block of a tile.  A zero-initialised state makes the first iteration of a fresh slot exact without a special case:
//     the first variable->check message is R * -1 + 1.0 * Q = -0.0 + Q = Q bit for bit (:199 vs :244-245).
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

constexpr int MAXG = 16;
constexpr size_t LDS_BYTES = 160 * 1024;

struct ResParams {
    double *llr;             // [B][n_v], clipped in place (:186)
    double *stage;           // [B][n_v] a-posteriori LLRs of retired blocks
    int32_t *iters;          // [B] executed iterations, may be null
    int *queue;              // next block to hand out
    const int32_t *row_ptr, *row_pad, *col_ptr, *col_pad;   // col_pad: (check << 5 | position) min-sum, edge id sum-product
    int64_t B, E;
    int n_v, nvp, n_c, cpad, vpad, max_iter;
    int dbg;
};

struct SlotTab { int blk[MAXG], k[MAXG], flag[MAXG], oblk[MAXG], ok[MAXG]; };

// ---- min-sum check pass of one (check, slot): syndrome bit + new record (:203-206, :229-238, :244-245) ----------
template <int G>
__device__ __forceinline__ void check_msa(const ResParams &p, const double *__restrict__ Q, double2 *__restrict__ M,
                                          uint2 *__restrict__ T, int c, int cw, int *flag) {
    const int cq = (p.dbg & 1) ? (c & 63) : c;
    const int deg = p.row_ptr[cq + 1] - p.row_ptr[cq];
    const int32_t *__restrict__ ev = p.row_pad + (int64_t)((p.dbg & 1) ? (c & 63) : c) * p.cpad;
    const double2 om = M[c * G + cw];
    const uint2 ot = T[c * G + cw];
    const MsaRec o{om.x, om.y, ot.y, (int)(ot.x & 0xffu), (int)((ot.x >> 8) & 1u)};
    int sx = 0, imin = 0;
    unsigned neg = 0;
    double m1 = __builtin_huge_val(), m2 = __builtin_huge_val();
    for (int j0 = 0; __builtin_amdgcn_ballot_w64(j0 < deg) != 0; j0 += 4) {      // rows are padded to a multiple of 4
        const int4 e = *reinterpret_cast<const int4 *>(ev + j0);
        const double q[4] = {Q[e.x * G + cw], Q[e.y * G + cw], Q[e.z * G + cw], Q[e.w * G + cw]};
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + u;
            if (j < deg) {
                sx ^= __double2hiint(q[u]);                      // dec_word = out_llrs < 0 (:193, :248)
                const double m = msa_edge(o, j, 1) + q[u];       // data * -1 + 1.0 * (msg_sum + llr) (:244-245); first pass: -0.0 + q (:199)
                const double a = fabs(m);
                const bool c1 = a < m1;
                m2 = min_f64(m2, c1 ? m1 : a);
                m1 = min_f64(m1, a);
                imin = c1 ? j : imin;
                neg |= (m < 0.0) ? (1u << j) : 0u;
            }
        }
    }
    if (sx < 0) *flag = 1;                                       // odd row: this iteration is executed (:205)
    M[c * G + cw] = double2{m1, m2};
    T[c * G + cw] = uint2{(unsigned)imin | ((unsigned)(__popc(neg) & 1) << 8), neg};
}

// ---- min-sum variable pass of one (variable, slot): column sum in increasing check order + llr (:243-247) ----------
template <int G>
__device__ __forceinline__ void var_msa(const ResParams &p, double *__restrict__ Q, const double2 *__restrict__ M,
                                        const uint2 *__restrict__ T, int v, int cw, const double *__restrict__ lrow) {
    const int vq = (p.dbg & 1) ? (v & 63) : v;
    const int deg = p.col_ptr[vq + 1] - p.col_ptr[vq];
    const int32_t *__restrict__ refs = p.col_pad + (int64_t)((p.dbg & 1) ? (v & 63) : v) * p.vpad;
    const double l = (p.dbg & 2) ? 1.0 : lrow[v];
    double msum = 0.0;
    for (int q0 = 0; __builtin_amdgcn_ballot_w64(q0 < deg) != 0; q0 += 4) {
        const int4 r4 = *reinterpret_cast<const int4 *>(refs + q0);
        const int ref[4] = {r4.x, r4.y, r4.z, r4.w};
        double2 mm[4];
        uint2 tt[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            mm[u] = M[(ref[u] >> 5) * G + cw];
            tt[u] = T[(ref[u] >> 5) * G + cw];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (q0 + u < deg) {
                const int j = ref[u] & 31;
                const double mn = (j == (int)(tt[u].x & 0xffu)) ? mm[u].y : mm[u].x;
                const unsigned ng = ((tt[u].y >> j) ^ (tt[u].x >> 8)) & 1u;
                msum += __hiloint2double(__double2hiint(mn) | (int)(ng << 31), __double2loint(mn));
            }
        }
    }
    Q[v * G + cw] = msum + l;                                    // msg_sum + llr (:245, :247)
}

// ---- sum-product check pass (:209-227); the tanh values of the row are parked in the row's own R entries ----------
template <int G>
__device__ __forceinline__ void check_spa(const ResParams &p, const double *__restrict__ Q, double *__restrict__ R, int c,
                                          int cw, int *flag) {
    const int e0 = p.row_ptr[c], deg = p.row_ptr[c + 1] - e0;
    const int32_t *__restrict__ ev = p.row_pad + (int64_t)((p.dbg & 1) ? (c & 63) : c) * p.cpad;
    double *__restrict__ Rr = R + (int64_t)e0 * G + cw;
    int sx = 0;
    double prod = 1.0;
    for (int j0 = 0; __builtin_amdgcn_ballot_w64(j0 < deg) != 0; j0 += 4) {
        const int4 e = *reinterpret_cast<const int4 *>(ev + j0);
        const double q[4] = {Q[e.x * G + cw], Q[e.y * G + cw], Q[e.z * G + cw], Q[e.w * G + cw]};
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + u;
            if (j < deg) {
                sx ^= __double2hiint(q[u]);                      // dec_word = out_llrs < 0 (:193, :248)
                double m = Rr[j * G] * -1.0;                     // data *= -1 (:244); first pass: 0 * -1 = -0.0
                m += 1.0 * q[u];                                 // data += H.multiply(msg_sum + llr).data (:245); first pass (:199)
                m = tanh_half(m);                                // data *= .5; tanh (:210-211)
                prod *= m;                                       // row product (reference: exp2(sum(log2)) :217-219)
                Rr[j * G] = m;
            }
        }
    }
    if (sx < 0) *flag = 1;
    for (int j = 0; __builtin_amdgcn_ballot_w64(j < deg) != 0; j++) {
        if (j < deg) {
            double x = (1.0 / Rr[j * G]) * prod;                 // data = 1/data; multiply(msg_products) (:222-223)
            x = clip_nan(x, -1.0, 1.0);                          // (:224)
            x = atanh_twice(x);                                  // (:225-226)
            Rr[j * G] = clip_nan(x, -500.0, 500.0);              // (:227)
        }
    }
}

template <int G>
__device__ __forceinline__ void var_spa(const ResParams &p, double *__restrict__ Q, const double *__restrict__ R, int v,
                                        int cw, const double *__restrict__ lrow) {
    const int vq = (p.dbg & 1) ? (v & 63) : v;
    const int deg = p.col_ptr[vq + 1] - p.col_ptr[vq];
    const int32_t *__restrict__ refs = p.col_pad + (int64_t)((p.dbg & 1) ? (v & 63) : v) * p.vpad;
    const double l = (p.dbg & 2) ? 1.0 : lrow[v];
    double msum = 0.0;
    for (int q0 = 0; __builtin_amdgcn_ballot_w64(q0 < deg) != 0; q0 += 4) {
        const int4 r4 = *reinterpret_cast<const int4 *>(refs + q0);
        const double r[4] = {R[(int64_t)r4.x * G + cw], R[(int64_t)r4.y * G + cw], R[(int64_t)r4.z * G + cw],
                             R[(int64_t)r4.w * G + cw]};
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (q0 + u < deg) msum += r[u];                      // message_matrix.sum(0) in increasing check order (:243)
    }
    Q[v * G + cw] = msum + l;
}

template <int ALG, int LG>
__global__ __launch_bounds__(1024) void ldpc_resident_kernel(ResParams p) {
    extern __shared__ __align__(16) double lds[];
    constexpr int G = 1 << LG;
    double *__restrict__ Q = lds;                                             // [nvp][G]
    double *__restrict__ R = Q + (int64_t)p.nvp * G;                          // sum-product: [E][G]
    double2 *__restrict__ M = reinterpret_cast<double2 *>(R);                 // min-sum: (min1, min2) [n_c][G]
    uint2 *__restrict__ T = reinterpret_cast<uint2 *>(M + (int64_t)p.n_c * G);   // min-sum: (imin | parity << 8, negatives)
    SlotTab *sl = ALG == CPX_LDPC_MSA ? reinterpret_cast<SlotTab *>(T + (int64_t)p.n_c * G)
                                      : reinterpret_cast<SlotTab *>(R + p.E * G);
    const int tid = threadIdx.x, cw = tid & (G - 1), sub = tid >> LG, npar = blockDim.x >> LG;
    if (tid < G) { sl->blk[tid] = -1; sl->k[tid] = 0; sl->flag[tid] = 0; }
    bool drained = false;                                                     // of the slot's own thread (tid < G)
    for (;;) {
        // ---- slots: an empty slot, or one that has used all its iterations, takes the next block of the queue ----
        if (tid < G) {
            const int b = sl->blk[tid], k = sl->k[tid];
            sl->oblk[tid] = b;
            sl->ok[tid] = k;
            if (b < 0 || k >= p.max_iter) {
                int nb = -1;
                if (!drained) {
                    const int t = atomicAdd(p.queue, 1);
                    if (t < p.B) nb = t; else drained = true;
                }
                sl->blk[tid] = nb;
                sl->k[tid] = 0;
            }
        }
        __syncthreads();
        const int b_old = sl->oblk[cw], b = sl->blk[cw];
        if (b_old != b) {                                                     // block indices are unique: replaced
            if (b_old >= 0) {                                                 // out of iterations: retire as it is
                double *__restrict__ out = p.stage + (int64_t)b_old * p.n_v;
                for (int v = sub; v < p.n_v; v += npar) out[v] = Q[v * G + cw];
                if (sub == 0 && p.iters) p.iters[b_old] = sl->ok[cw];
            }
            if (b >= 0) {
                double *__restrict__ in = p.llr + (int64_t)b * p.n_v;
                for (int v = sub; v < p.n_v; v += npar) {
                    const double raw = in[v];
                    const double x = clip_nan(raw, -500.0, 500.0);
                    if (x != raw) in[v] = x;                                  // in-place clip (:186); untouched values are not rewritten
                    Q[v * G + cw] = x;                                        // out_llrs = llr (:194)
                }
                if (ALG == CPX_LDPC_MSA) {
                    for (int c = sub; c < p.n_c; c += npar) { M[c * G + cw] = double2{0.0, 0.0}; T[c * G + cw] = uint2{0u, 0u}; }
                } else {
                    for (int64_t e = sub; e < p.E; e += npar) R[e * G + cw] = 0.0;
                }
            }
        }
        int any = 0, rep = 0;
#pragma unroll
        for (int g = 0; g < G; g++) { any |= sl->blk[g] >= 0; rep |= sl->blk[g] != sl->oblk[g]; }
        if (!any) break;
        if (rep) __syncthreads();
        // ---- check pass ----
        if (b >= 0) {
            for (int c = sub; c < p.n_c; c += npar) {
                if (ALG == CPX_LDPC_MSA) check_msa<G>(p, Q, M, T, c, cw, &sl->flag[cw]);
                else check_spa<G>(p, Q, R, c, cw, &sl->flag[cw]);
            }
        }
        __syncthreads();
        // ---- verdict: an unsatisfied slot runs the variable pass, a satisfied one is retired with the Q it has (:205-206) ----
        if (b >= 0) {
            if (sl->flag[cw]) {
                const double *__restrict__ lrow = p.llr + (int64_t)b * p.n_v;
                for (int v = sub; v < p.n_v; v += npar) {
                    if (ALG == CPX_LDPC_MSA) var_msa<G>(p, Q, M, T, v, cw, lrow);
                    else var_spa<G>(p, Q, R, v, cw, lrow);
                }
            } else {
                double *__restrict__ out = p.stage + (int64_t)b * p.n_v;
                for (int v = sub; v < p.n_v; v += npar) out[v] = Q[v * G + cw];
                if (sub == 0 && p.iters) p.iters[b] = sl->k[cw];
            }
        }
        __syncthreads();
        if (tid < G && sl->blk[tid] >= 0) {
            if (sl->flag[tid]) { sl->k[tid] += 1; sl->flag[tid] = 0; }
            else sl->blk[tid] = -1;
        }
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

size_t state_doubles(const cpx_ldpc *c, int alg, int G) {
    const size_t nvp = (size_t)(c->n_v + 1) & ~(size_t)1;
    return alg == CPX_LDPC_MSA ? (nvp + 3 * (size_t)c->n_c) * G : (nvp + (size_t)c->n_edges) * G;
}

// workgroup size: the multiple of 64 in [512, 1024] that wastes the fewest thread slots in the last round of the
// two passes (thread = (node, slot); a pass is ceil(nodes / (threads / G)) rounds)
int pick_threads(const cpx_ldpc *c, int G) {
    const double wc = 40.0 + 14.0 * (double)c->n_edges / c->n_c, wv = 25.0 + 12.0 * (double)c->n_edges / c->n_v;
    int best = 1024;
    double best_cost = 1e300;
    for (int n = 1024; n >= 512; n -= 64) {
        const int npar = n / G;
        const double cost = (double)n * (((c->n_c + npar - 1) / npar) * wc + ((c->n_v + npar - 1) / npar) * wv);
        if (cost < best_cost * 0.97) { best_cost = cost; best = n; }         // larger workgroups win near-ties
    }
    return best;
}

std::atomic<int> g_ldpc_path{-1};                                 // 0 auto, 1 tiled, 2 resident (strict)
int parse_ldpc_path(const char *m) {
    if (!m || !m[0] || strcmp(m, "auto") == 0) return 0;
    if (strcmp(m, "tiled") == 0) return 1;
    if (strcmp(m, "resident") == 0) return 2;
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

template <int ALG, int LG>
int launch_resident(const ResParams &p, int grid, int threads, size_t lds, hipStream_t st) {
    static bool raised[64] = {};                                  // > 64 KiB of dynamic LDS is opt-in, once per kernel and device
    static std::mutex raised_mu;
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lk(raised_mu);
        if (dev >= 0 && dev < 64 && !raised[dev]) {
            CPX_HIP(hipFuncSetAttribute((const void *)ldpc_resident_kernel<ALG, LG>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)LDS_BYTES));
            raised[dev] = true;
        }
    }
    hipLaunchKernelGGL((ldpc_resident_kernel<ALG, LG>), dim3((unsigned)grid), dim3((unsigned)threads), lds, st, p);
    return CPX_OK;
}

}  // namespace

namespace cpx {

bool ldpc_resident_path(const cpx_ldpc *c, double *d_llr, int64_t B, int alg, int n_iters, int8_t *d_dec, double *d_out,
                        int32_t *d_iters, hipStream_t st, int *rc) {
    *rc = CPX_OK;
    const int mode = ldpc_path();
    auto reject = [&](const char *why) {
        if (mode == 2) { set_error("ldpc: resident path forced but not applicable: %s", why); *rc = CPX_EINVAL; return true; }
        return false;
    };
    if (mode == 1) return false;
    if (n_iters < 1) return reject("n_iters == 0");
    if (B >= (1ll << 30)) return reject("batch too large");
    int LG = -1;
    for (int lg = 4; lg >= 0; lg--)
        if (8 * state_doubles(c, alg, 1 << lg) + sizeof(SlotTab) + 64 <= LDS_BYTES) { LG = lg; break; }
    if (const char *e = getenv("CPX_LDPC_G")) {                   // experiment knob: slots per workgroup (log2)
        const int lg = atoi(e);
        if (lg >= 0 && lg <= LG) LG = lg;
    }
    if (LG < 0) return reject("decoder state of one block exceeds the LDS of a compute unit");
    const int G = 1 << LG;
    int threads = pick_threads(c, G);
    if (const char *e = getenv("CPX_LDPC_THREADS")) {             // experiment knob
        const int t = atoi(e);
        if (t >= 64 && t <= 1024 && t % 64 == 0) threads = t;
    }
    const size_t lds = 8 * state_doubles(c, alg, G) + sizeof(SlotTab) + 64;
    char *slab = nullptr;
    const size_t sz_stage = (sizeof(double) * (size_t)(B * c->n_v) + 255) & ~(size_t)255;
    if ((*rc = workspace(st, 0, sz_stage + 256, (void **)&slab))) return true;
    ResParams p;
    p.llr = d_llr; p.stage = (double *)slab; p.iters = d_iters; p.queue = (int *)(slab + sz_stage);
    p.row_ptr = c->d_row_ptr; p.row_pad = c->d_row_pad; p.col_ptr = c->d_col_ptr;
    p.col_pad = alg == CPX_LDPC_MSA ? c->d_col_pad_cj : c->d_col_pad_edge;
    p.B = B; p.E = c->n_edges; p.n_v = c->n_v; p.nvp = (c->n_v + 1) & ~1; p.n_c = c->n_c; p.cpad = c->cpad; p.vpad = c->vpad;
    p.max_iter = n_iters;
    p.dbg = getenv("CPX_LDPC_DBG") ? atoi(getenv("CPX_LDPC_DBG")) : 0;
    if (hipMemsetAsync(p.queue, 0, sizeof(int), st) != hipSuccess) { set_error("ldpc: hipMemsetAsync failed"); *rc = CPX_EHIP; return true; }
    // one workgroup per compute unit (its LDS holds one set of slots), no more than there are slot sets to fill
    const int per_cu = std::max<int>(1, (int)(LDS_BYTES / lds));
    const int grid = (int)std::min<int64_t>((int64_t)device_cus() * std::min(per_cu, std::max(1, 2048 / threads)), (B + G - 1) / G);
    int lrc = CPX_OK;
#define CPX_RES(A, L) case (A) * 8 + (L): lrc = launch_resident<A, L>(p, grid, threads, lds, st); break;
    switch (alg * 8 + LG) {
        CPX_RES(CPX_LDPC_SPA, 0) CPX_RES(CPX_LDPC_SPA, 1) CPX_RES(CPX_LDPC_SPA, 2) CPX_RES(CPX_LDPC_SPA, 3) CPX_RES(CPX_LDPC_SPA, 4)
        CPX_RES(CPX_LDPC_MSA, 0) CPX_RES(CPX_LDPC_MSA, 1) CPX_RES(CPX_LDPC_MSA, 2) CPX_RES(CPX_LDPC_MSA, 3) CPX_RES(CPX_LDPC_MSA, 4)
    }
#undef CPX_RES
    if (lrc) { *rc = lrc; return true; }
    hipLaunchKernelGGL(ldpc_unstage_kernel, dim3((unsigned)((B + 63) / 64), (unsigned)((c->n_v + 63) / 64)), dim3(256), 0, st,
                       p.stage, B, c->n_v, d_out, d_dec);
    if (hipGetLastError() != hipSuccess) { set_error("ldpc (resident path): launch failed"); *rc = CPX_EHIP; }
    note_kernel("ldpc_resident_kernel<%s,G=%d,threads=%d>", alg == CPX_LDPC_MSA ? "MSA" : "SPA", G, threads);
    return true;
}

}  // namespace cpx

extern "C" int cpx_ldpc_set_path(const char *mode) {
    const int v = parse_ldpc_path(mode);
    if (v < 0) {
        cpx::set_error("cpx_ldpc_set_path: unknown mode '%s' (auto | tiled | resident)", mode);
        return CPX_EINVAL;
    }
    g_ldpc_path.store(v, std::memory_order_relaxed);
    return CPX_OK;
}
