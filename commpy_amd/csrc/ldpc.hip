// LDPC belief propagation (flooding schedule, SPA and min-sum) for gfx950.  Replaces the body of
//   ldpc_bp_decode  (/root/reference/commpy/channelcoding/ldpc.py:144-254)
// for a batch of B independent blocks (the reference's sequential `for i_start` loop, ldpc.py:197).
//
// Data layout in HBM (codeword index fastest => every access below is coalesced across lanes):
//   M      [E][B]  float64 edge messages, edges sorted by (check, variable)  -- message_matrix
//   llrT   [n_v][B] float64 clipped channel LLRs (transposed once from the caller's [B][n_v])
//   out    [n_v][B] float64 / dec [n_v][B] int8 -- exactly the reference's output layout (:251-253)
//   unsat  [B] int32 "syndrome not yet zero" flag, recomputed at the start of every iteration
// One iteration = syndrome pass (early-exit test, :205-206) + check-node pass + variable-node pass;
// a block whose syndrome is zero is frozen: its threads exit before touching messages, which
// reproduces the reference's `break` per block.  Lanes = consecutive codewords; a thread owns one
// (check, codeword) or (variable, codeword) pair and reduces its row/column in registers in the
// reference's order (row: increasing variable; column: increasing check -- SciPy coo_matvec order).
// HBM traffic per executed iteration per block = (4E + 2 n_v) * 8 B (algorithmic).
#include "cpx_internal.h"

#include <algorithm>

using namespace cpx;

namespace {

constexpr int LB = 256;     // threads per block (codewords along x)
constexpr int MAXDEG = 32;  // max check degree held in registers

__device__ __forceinline__ double clip_nan(double v, double lo, double hi) {
    // np.clip propagates NaN
    return (v != v) ? v : fmin(fmax(v, lo), hi);
}

// llrT[v][b] = clip(llr[b][v]); llr clipped in place (ldpc.py:186); out = llr; dec = signbit (:193-194)
__global__ __launch_bounds__(LB) void ldpc_init_kernel(double *__restrict__ llr, int64_t B, int n_v,
                                                       double *__restrict__ llrT, double *__restrict__ out,
                                                       int8_t *__restrict__ dec, int32_t *__restrict__ iters,
                                                       int32_t *__restrict__ unsat) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const int64_t b0 = (int64_t)blockIdx.x * 32;
    const int v0 = blockIdx.y * 32;
    for (int r = ty; r < 32; r += 8) {                            // read rows = codewords, cols = variables
        const int64_t b = b0 + r;
        const int v = v0 + tx;
        double x = 0.0;
        if (b < B && v < n_v) {
            x = clip_nan(llr[b * n_v + v], -500.0, 500.0);
            llr[b * n_v + v] = x;
        }
        tile[r][tx] = x;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {                            // write rows = variables, cols = codewords
        const int v = v0 + r;
        const int64_t b = b0 + tx;
        if (b < B && v < n_v) {
            const double x = tile[tx][r];
            llrT[(int64_t)v * B + b] = x;
            out[(int64_t)v * B + b] = x;
            dec[(int64_t)v * B + b] = (int8_t)(__builtin_signbit(x) ? 1 : 0);
        }
    }
    if (blockIdx.y == 0) {
        const int64_t b = b0 + threadIdx.x;
        if (threadIdx.x < 32 && b < B) {
            if (iters) iters[b] = 0;
            unsat[b] = 0;
        }
    }
}

// message_matrix = H.multiply(llr) (:199): M[e][b] = llr[var(e)][b]
__global__ __launch_bounds__(LB) void ldpc_msg_init_kernel(const double *__restrict__ llrT, int64_t B,
                                                           const int32_t *__restrict__ edge_var,
                                                           int64_t E, double *__restrict__ M) {
    const int64_t b = (int64_t)blockIdx.x * LB + threadIdx.x;
    if (b >= B) return;
    for (int64_t e = blockIdx.y; e < E; e += gridDim.y) M[e * B + b] = 1.0 * llrT[(int64_t)edge_var[e] * B + b];
}

// Early-termination test (:205): unsat[b] == stamp <=> some check of block b has odd parity of dec_word.
__global__ __launch_bounds__(LB) void ldpc_syndrome_kernel(const int8_t *__restrict__ dec, int64_t B,
                                                           const int32_t *__restrict__ row_ptr,
                                                           const int32_t *__restrict__ edge_var,
                                                           int32_t *__restrict__ unsat, int stamp) {
    const int64_t b = (int64_t)blockIdx.x * LB + threadIdx.x;
    const int c = blockIdx.y;
    if (b >= B) return;
    int par = 0;
    for (int e = row_ptr[c]; e < row_ptr[c + 1]; e++) par ^= dec[(int64_t)edge_var[e] * B + b];
    if (par & 1) unsat[b] = stamp;      // stamp = iteration + 1: no per-iteration clearing pass needed
}

// Check-node update.  SPA (:209-227) / MSA (:229-238).
template <int ALG, int DEG_CAP>
__global__ __launch_bounds__(LB) void ldpc_cn_kernel(double *__restrict__ M, int64_t B,
                                                     const int32_t *__restrict__ row_ptr,
                                                     const int32_t *__restrict__ unsat, int stamp) {
    const int64_t b = (int64_t)blockIdx.x * LB + threadIdx.x;
    const int c = blockIdx.y;
    if (b >= B || unsat[b] != stamp) return;
    const int e0 = row_ptr[c];
    const int deg = row_ptr[c + 1] - e0;
    double v[DEG_CAP];
#pragma unroll
    for (int j = 0; j < DEG_CAP; j++) v[j] = (j < deg) ? M[(int64_t)(e0 + j) * B + b] : 1.0;
    if (ALG == CPX_LDPC_SPA) {
        double prod = 1.0;
#pragma unroll
        for (int j = 0; j < DEG_CAP; j++) {
            if (j < deg) {
                v[j] = tanh(v[j] * 0.5);                 // data *= .5; tanh (:210-211)
                prod *= v[j];                            // row product (reference: exp2(sum(log2)) :217-219), increasing variable
            }
        }
#pragma unroll
        for (int j = 0; j < DEG_CAP; j++) {
            if (j < deg) {
                double x = (1.0 / v[j]) * prod;          // data = 1/data; multiply(msg_products) (:222-223)
                x = clip_nan(x, -1.0, 1.0);              // (:224)
                x = atanh(x) * 2.0;                      // (:225-226)
                M[(int64_t)(e0 + j) * B + b] = clip_nan(x, -500.0, 500.0);   // (:227)
            }
        }
    } else {
        // sign(other).prod() * abs(other).min() over the OTHER edges of the row (:236-237)
        int nzero = 0, nneg = 0, imin = -1;
        double min1 = __builtin_huge_val(), min2 = __builtin_huge_val();
#pragma unroll
        for (int j = 0; j < DEG_CAP; j++) {
            if (j < deg) {
                const double a = fabs(v[j]);
                nzero += (v[j] == 0.0);
                nneg += (v[j] < 0.0);
                if (a < min1) { min2 = min1; min1 = a; imin = j; }
                else if (a < min2) { min2 = a; }
            }
        }
#pragma unroll
        for (int j = 0; j < DEG_CAP; j++) {
            if (j < deg) {
                const int oz = nzero - (v[j] == 0.0);
                const int on = nneg - (v[j] < 0.0);
                const double sgn = oz ? 0.0 : ((on & 1) ? -1.0 : 1.0);
                const double mn = (j == imin) ? min2 : min1;
                M[(int64_t)(e0 + j) * B + b] = sgn * mn;
            }
        }
    }
}

// Variable-node update (:243-248).
__global__ __launch_bounds__(LB) void ldpc_vn_kernel(double *__restrict__ M, int64_t B,
                                                     const int32_t *__restrict__ col_ptr,
                                                     const int32_t *__restrict__ col_edge,
                                                     const double *__restrict__ llrT, double *__restrict__ out,
                                                     int8_t *__restrict__ dec, const int32_t *__restrict__ unsat,
                                                     int stamp, int32_t *__restrict__ iters) {
    const int64_t b = (int64_t)blockIdx.x * LB + threadIdx.x;
    const int v = blockIdx.y;
    if (b >= B || unsat[b] != stamp) return;
    const int q0 = col_ptr[v], q1 = col_ptr[v + 1];
    double msum = 0.0;                                           // message_matrix.sum(0): increasing check
    for (int q = q0; q < q1; q++) msum += M[(int64_t)col_edge[q] * B + b];
    const double tot = msum + llrT[(int64_t)v * B + b];          // msg_sum + llr (:245, :247)
    for (int q = q0; q < q1; q++) {
        const int64_t idx = (int64_t)col_edge[q] * B + b;
        double m = M[idx] * -1.0;                                // data *= -1 (:244)
        m += 1.0 * tot;                                          // data += H.multiply(msg_sum + llr).data (:245)
        M[idx] = m;
    }
    out[(int64_t)v * B + b] = tot;                               // (:247)
    dec[(int64_t)v * B + b] = (int8_t)(__builtin_signbit(tot) ? 1 : 0);   // (:248)
    if (v == 0 && iters) iters[b] += 1;
}

}  // namespace

extern "C" {

int cpx_ldpc_create(int n_vnodes, int n_cnodes, int64_t n_edges, const int32_t *edge_check, const int32_t *edge_var,
                    cpx_ldpc **out) {
    CPX_REQUIRE(out && edge_check && edge_var, CPX_EINVAL, "cpx_ldpc_create: null pointer");
    CPX_REQUIRE(n_vnodes > 0 && n_cnodes > 0 && n_edges > 0, CPX_EINVAL, "cpx_ldpc_create: empty code");
    CPX_REQUIRE(n_edges < (1ll << 30), CPX_ELIMIT, "cpx_ldpc_create: too many edges");
    int rc = ensure_device();
    if (rc) return rc;
    const int64_t E = n_edges;
    std::vector<int32_t> row_ptr(n_cnodes + 1, 0), col_ptr(n_vnodes + 1, 0), col_edge(E);
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
    CPX_REQUIRE(n_vnodes <= 65535 * 32 && n_cnodes <= 65535, CPX_ELIMIT, "cpx_ldpc_create: code too large (n_c <= 65535)");
    CPX_REQUIRE(n_vnodes <= 65535, CPX_ELIMIT, "cpx_ldpc_create: code too large (n_v <= 65535)");
    CPX_REQUIRE(max_cdeg <= MAXDEG, CPX_ELIMIT, "cpx_ldpc_create: check degree %d > %d not supported", max_cdeg, MAXDEG);
    std::vector<int32_t> fill(col_ptr.begin(), col_ptr.end() - 1);
    for (int64_t e = 0; e < E; e++) col_edge[fill[edge_var[e]]++] = (int32_t)e;   // increasing e == increasing check
    cpx_ldpc *c = new cpx_ldpc;
    c->n_v = n_vnodes; c->n_c = n_cnodes; c->n_edges = E; c->max_cdeg = max_cdeg; c->max_vdeg = max_vdeg;
    (void)hipGetDevice(&c->device);
    CPX_HIP(hipMalloc((void **)&c->d_edge_var, sizeof(int32_t) * E));
    CPX_HIP(hipMalloc((void **)&c->d_row_ptr, sizeof(int32_t) * (n_cnodes + 1)));
    CPX_HIP(hipMalloc((void **)&c->d_col_ptr, sizeof(int32_t) * (n_vnodes + 1)));
    CPX_HIP(hipMalloc((void **)&c->d_col_edge, sizeof(int32_t) * E));
    CPX_HIP(hipMemcpy(c->d_edge_var, edge_var, sizeof(int32_t) * E, hipMemcpyHostToDevice));
    CPX_HIP(hipMemcpy(c->d_row_ptr, row_ptr.data(), sizeof(int32_t) * (n_cnodes + 1), hipMemcpyHostToDevice));
    CPX_HIP(hipMemcpy(c->d_col_ptr, col_ptr.data(), sizeof(int32_t) * (n_vnodes + 1), hipMemcpyHostToDevice));
    CPX_HIP(hipMemcpy(c->d_col_edge, col_edge.data(), sizeof(int32_t) * E, hipMemcpyHostToDevice));
    *out = c;
    return CPX_OK;
}

int cpx_ldpc_destroy(cpx_ldpc *c) {
    if (!c) return CPX_OK;
    (void)hipFree(c->d_edge_var); (void)hipFree(c->d_row_ptr); (void)hipFree(c->d_col_ptr); (void)hipFree(c->d_col_edge);
    delete c;
    return CPX_OK;
}

int cpx_ldpc_bp_decode_batch_dev(const cpx_ldpc *c, double *d_llr, int64_t B, int alg, int n_iters, int8_t *d_dec,
                                 double *d_out, int32_t *d_iters, void *stream) {
    CPX_REQUIRE(c, CPX_EINVAL, "ldpc: null code");
    CPX_REQUIRE(alg == CPX_LDPC_SPA || alg == CPX_LDPC_MSA, CPX_EINVAL,
                "Please input a valid decoder_algorithm string (meanning \"SPA\" or \"MSA\").");
    CPX_REQUIRE(B >= 0 && n_iters >= 0, CPX_EINVAL, "ldpc: negative size");
    if (B == 0) return CPX_OK;
    hipStream_t st = pick_stream(stream);
    const int64_t E = c->n_edges;
    double *M = nullptr, *llrT = nullptr;
    int32_t *unsat = nullptr;
    int rcw;
    if ((rcw = workspace(st, 0, sizeof(double) * (size_t)(E * B), (void **)&M))) return rcw;
    if ((rcw = workspace(st, 1, sizeof(double) * (size_t)((int64_t)c->n_v * B), (void **)&llrT))) return rcw;
    if ((rcw = workspace(st, 2, sizeof(int32_t) * (size_t)B, (void **)&unsat))) return rcw;
    const unsigned bx = (unsigned)((B + LB - 1) / LB);
    {
        dim3 grid((unsigned)((B + 31) / 32), (unsigned)((c->n_v + 31) / 32));
        hipLaunchKernelGGL(ldpc_init_kernel, grid, dim3(LB), 0, st, d_llr, B, c->n_v, llrT, d_out, d_dec, d_iters, unsat);
        hipLaunchKernelGGL(ldpc_msg_init_kernel, dim3(bx, (unsigned)std::min<int64_t>(E, 65535)), dim3(LB), 0, st, llrT, B,
                           c->d_edge_var, E, M);
    }
    for (int it = 0; it < n_iters; it++) {
        const int stamp = it + 1;
        hipLaunchKernelGGL(ldpc_syndrome_kernel, dim3(bx, (unsigned)c->n_c), dim3(LB), 0, st, d_dec, B, c->d_row_ptr,
                           c->d_edge_var, unsat, stamp);
        dim3 gc(bx, (unsigned)c->n_c);
#define LAUNCH_CN(ALG)                                                                                              \
    if (c->max_cdeg <= 8) hipLaunchKernelGGL((ldpc_cn_kernel<ALG, 8>), gc, dim3(LB), 0, st, M, B, c->d_row_ptr, unsat, stamp); \
    else if (c->max_cdeg <= 16) hipLaunchKernelGGL((ldpc_cn_kernel<ALG, 16>), gc, dim3(LB), 0, st, M, B, c->d_row_ptr, unsat, stamp); \
    else hipLaunchKernelGGL((ldpc_cn_kernel<ALG, MAXDEG>), gc, dim3(LB), 0, st, M, B, c->d_row_ptr, unsat, stamp);
        if (alg == CPX_LDPC_SPA) { LAUNCH_CN(CPX_LDPC_SPA) } else { LAUNCH_CN(CPX_LDPC_MSA) }
#undef LAUNCH_CN
        hipLaunchKernelGGL(ldpc_vn_kernel, dim3(bx, (unsigned)c->n_v), dim3(LB), 0, st, M, B, c->d_col_ptr,
                           c->d_col_edge, llrT, d_out, d_dec, unsat, stamp, d_iters);
    }
    CPX_HIP(hipGetLastError());
    return CPX_OK;
}

int cpx_ldpc_bp_decode_batch(const cpx_ldpc *c, double *llr, int64_t B, int alg, int n_iters, int8_t *dec_word,
                             double *out_llrs, int32_t *iters_done) {
    CPX_REQUIRE(c && (llr || B == 0), CPX_EINVAL, "ldpc: null pointer");
    int rc = ensure_device();
    if (rc) return rc;
    if (B == 0) return CPX_OK;
    const size_t nvb = (size_t)((int64_t)c->n_v * B);
    DevBuf d_llr, d_dec, d_out, d_it;
    if ((rc = d_llr.alloc(sizeof(double) * nvb))) return rc;
    if ((rc = d_dec.alloc(nvb))) return rc;
    if ((rc = d_out.alloc(sizeof(double) * nvb))) return rc;
    if ((rc = d_it.alloc(sizeof(int32_t) * (size_t)B))) return rc;
    hipStream_t st = lib_stream();
    CPX_HIP(hipMemcpyAsync(d_llr.p, llr, sizeof(double) * nvb, hipMemcpyHostToDevice, st));
    rc = cpx_ldpc_bp_decode_batch_dev(c, d_llr.as<double>(), B, alg, n_iters, d_dec.as<int8_t>(), d_out.as<double>(),
                                      d_it.as<int32_t>(), st);
    if (rc) return rc;
    CPX_HIP(hipMemcpyAsync(llr, d_llr.p, sizeof(double) * nvb, hipMemcpyDeviceToHost, st));   // in-place clip (:186)
    if (dec_word) CPX_HIP(hipMemcpyAsync(dec_word, d_dec.p, nvb, hipMemcpyDeviceToHost, st));
    if (out_llrs) CPX_HIP(hipMemcpyAsync(out_llrs, d_out.p, sizeof(double) * nvb, hipMemcpyDeviceToHost, st));
    if (iters_done) CPX_HIP(hipMemcpyAsync(iters_done, d_it.p, sizeof(int32_t) * (size_t)B, hipMemcpyDeviceToHost, st));
    CPX_HIP(hipStreamSynchronize(st));
    return CPX_OK;
}

}  // extern "C"
