// BCJR / MAP decoder and fused turbo decoder for gfx950.  Replaces the bodies of
//   map_decode    (/root/reference/commpy/channelcoding/turbo.py:163-251)
//     _backward_recursion (:78-111), _forward_recursion_decoding (:114-158), _compute_branch_prob (:62-76)
//   turbo_decode  (turbo.py:254-333) with interlv/deinterlv (interleavers.py:13-47)
// Probability-domain recursions with per-step sum-normalisation, float64, exactly the reference's
// formulas (no log-MAP): gamma = exp(-((r0-c0)^2+(r1-c1)^2)/(2*nv)), priors p0 = 1/(1+e^L), p1 = 1-p0,
// beta_N = 1 (unterminated), alpha_0 = delta(state 0), L = L_int + log(app1/app0) (app without prior).
//
// Mapping (wave64): lane = g*S + s -- G = 64/S codewords per wavefront, one trellis state per lane.
//   * alpha / beta live in one VGPR pair per lane; neighbours (next states for beta, predecessors for
//     alpha) are fetched with wavefront shuffles; the sums over states (normalisation, app) are
//     xor-butterfly shuffle reductions inside the S-lane group;
//   * the 4 distinct branch probabilities of a step are computed once per group (lane s evaluates
//     code s&3) and shared by shuffle, so a step costs ONE exp per lane;
//   * priors are evaluated time-parallel (64 lanes over t) before the recursions;
//   * beta[t][s] (needed again by the forward pass) and the per-iteration L arrays of the turbo
//     loop stay in a per-codeword HBM scratch slab (L2/MALL resident at these sizes); the whole
//     iteration loop of turbo_decode runs inside ONE kernel launch, the interleaver being a
//     gather/scatter through that slab.
// Summation inside a group uses a butterfly instead of the reference's sequential order: the
// difference is O(1e-16) relative, far inside the 1e-5 parity tolerance for soft outputs.
#include "cpx_internal.h"

using namespace cpx;

namespace {

struct MapTables {
    const int32_t *next_state, *output;                       // [S][2]
    const int32_t *pred_state, *pred_input, *pred_code;       // [S][2]
    int lgS, n;
};

struct LaneCtx {
    int lane, lgS, S, G, g, s, gbase;
    int nxt[2], code[2];         // outgoing branches of state s: next-state lane, 2-bit code (sys,parity)
    int plane[2], pin[2], pcode[2];  // incoming branches in np.where order: predecessor lane, input, code
};

__device__ __forceinline__ double shf(double v, int src) { return __shfl(v, src, 64); }

__device__ __forceinline__ double group_sum(double v, int S) {
    for (int off = 1; off < S; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ void init_ctx(LaneCtx &c, const MapTables &tb) {
    c.lane = threadIdx.x;
    c.lgS = tb.lgS;
    c.S = 1 << tb.lgS;
    c.G = 64 >> tb.lgS;
    c.g = c.lane >> tb.lgS;
    c.s = c.lane & (c.S - 1);
    c.gbase = c.g << tb.lgS;
    const int sh = tb.n - 2;
    for (int i = 0; i < 2; i++) {
        c.nxt[i] = c.gbase + tb.next_state[c.s * 2 + i];
        c.code[i] = (tb.output[c.s * 2 + i] >> sh) & 3;       // [msg_bit, parity_bit] = codeword_array[0:2] (:96-98)
        c.plane[i] = c.gbase + tb.pred_state[c.s * 2 + i];
        c.pin[i] = tb.pred_input[c.s * 2 + i];
        c.pcode[i] = (tb.pred_code[c.s * 2 + i] >> sh) & 3;
    }
}

// branch probability of 2-bit code `code` (:62-76)
__device__ __forceinline__ double branch_prob(int code, double r0, double r1, double nv2) {
    const double c0 = (double)(2 * ((code >> 1) & 1) - 1);
    const double c1 = (double)(2 * (code & 1) - 1);
    const double x = r0 - c0, y = r1 - c1;
    return exp(-(x * x + y * y) / nv2);
}

// gamma of the two branches `codeA`, `codeB` needed by this lane; shared through the group when S >= 4.
__device__ __forceinline__ void gammas(const LaneCtx &c, double r0, double r1, double nv2, int codeA, int codeB,
                                       double &gA, double &gB) {
    if (c.S >= 4) {
        const double mine = branch_prob(c.s & 3, r0, r1, nv2);
        gA = shf(mine, c.gbase + codeA);
        gB = shf(mine, c.gbase + codeB);
    } else {
        gA = branch_prob(codeA, r0, r1, nv2);
        gB = branch_prob(codeB, r0, r1, nv2);
    }
}

// One MAP pass over the G codewords of this wavefront.
//   sys/par/Lin: per-lane base pointers of the lane's codeword (length N); sys is read through
//   `perm` when sys_perm != nullptr (sys_symbols_i = interlv(sys), turbo.py:310).
//   pr0/beta: scratch of the lane's codeword: pr0[N], beta[(N+1)*S].
//   Lout[N] receives L_int + log(app1/app0).
__device__ void map_pass(const LaneCtx &c, bool valid, int64_t N, double nv2, const double *sys, const int32_t *sys_perm,
                         const double *par, const double *Lin, double *pr0, double *beta, double *Lout) {
    const int S = c.S, G = c.G;
    // ---- priors, time-parallel: lane (g, s) handles t = s, s+S, ... of codeword g (:238-240) ----
    if (valid)
        for (int64_t t = c.s; t < N; t += S) pr0[t] = 1.0 / (1.0 + exp(Lin[t]));
    __syncthreads();
    // ---- backward recursion (:78-111) ----
    double b = 1.0;                                              // b_state_metrics[:, N] = 1 (:225)
    if (valid) beta[N * S + c.s] = b;
    for (int64_t t = N; t >= 1; --t) {
        double r0 = 0, r1 = 0, p0 = 0.5;
        if (valid) {
            r0 = sys[sys_perm ? sys_perm[t - 1] : (t - 1)];
            r1 = par[t - 1];
            p0 = pr0[t - 1];
        }
        const double p1 = 1.0 - p0;                              // priors[1] = 1 - priors[0] (:240)
        double g0, g1;
        gammas(c, r0, r1, nv2, c.code[0], c.code[1], g0, g1);
        const double bn0 = shf(b, c.nxt[0]), bn1 = shf(b, c.nxt[1]);
        double nb = 0.0;
        nb += (bn0 * g0 * p0);                                   // (:106-108), input 0 then input 1
        nb += (bn1 * g1 * p1);
        const double sum = group_sum(nb, S);
        b = nb / sum;                                            // (:110-111)
        if (valid) beta[(t - 1) * S + c.s] = b;
    }
    __syncthreads();
    // ---- forward recursion + LLR (:114-158) ----
    double a = (c.s == 0) ? 1.0 : 0.0;                           // f_state_metrics[0][0] = 1 (:221)
    for (int64_t t = 1; t <= N; ++t) {
        double r0 = 0, r1 = 0, p0 = 0.5, bt = 1.0, lin = 0.0;
        if (valid) {
            r0 = sys[sys_perm ? sys_perm[t - 1] : (t - 1)];
            r1 = par[t - 1];
            p0 = pr0[t - 1];
            bt = beta[t * S + c.s];
            lin = Lin[t - 1];
        }
        const double p1 = 1.0 - p0;
        double mine = 0.0, go0, go1, gi0, gi1;
        if (S >= 4) {
            mine = branch_prob(c.s & 3, r0, r1, nv2);
            go0 = shf(mine, c.gbase + c.code[0]);  go1 = shf(mine, c.gbase + c.code[1]);
            gi0 = shf(mine, c.gbase + c.pcode[0]); gi1 = shf(mine, c.gbase + c.pcode[1]);
        } else {
            go0 = branch_prob(c.code[0], r0, r1, nv2);  go1 = branch_prob(c.code[1], r0, r1, nv2);
            gi0 = branch_prob(c.pcode[0], r0, r1, nv2); gi1 = branch_prob(c.pcode[1], r0, r1, nv2);
        }
        // app[i] += f[cs,0] * branch_prob * b[next_state, t]   (:141-143)
        const double x0 = a * go0 * shf(bt, c.nxt[0]);
        const double x1 = a * go1 * shf(bt, c.nxt[1]);
        const double app0 = group_sum(x0, S), app1 = group_sum(x1, S);
        const double lappr = lin + log(app1 / app0);             // (:145)
        if (valid && c.s == 0) Lout[t - 1] = lappr;
        // f[next,1] += f[cs,0] * branch_prob * priors[input]     (:136-138), accumulation in (cs, input) order
        const double ap0 = shf(a, c.plane[0]), ap1 = shf(a, c.plane[1]);
        double na = 0.0;
        na += (ap0 * gi0 * (c.pin[0] ? p1 : p0));
        na += (ap1 * gi1 * (c.pin[1] ? p1 : p0));
        const double sum = group_sum(na, S);
        a = na / sum;                                            // (:155-158)
    }
    __syncthreads();
    (void)G;
}

struct MapParams {
    MapTables tb;
    const double *sys, *par, *Lin;     // [B][N]
    double *Lout;                      // [B][N]
    uint8_t *bits;                     // [B][N]
    double *scratch;                   // per codeword: pr0[N] + beta[(N+1)*S]
    int64_t B, N, slab;
    double nv2;
    int want_bits;
};

__global__ __launch_bounds__(64) void map_decode_kernel(MapParams p) {
    LaneCtx c;
    init_ctx(c, p.tb);
    const int64_t cw = (int64_t)blockIdx.x * c.G + c.g;
    const bool valid = cw < p.B;
    const int64_t o = (valid ? cw : 0) * p.N;
    double *slab = p.scratch + (valid ? cw : 0) * p.slab;
    map_pass(c, valid, p.N, p.nv2, p.sys + o, nullptr, p.par + o, p.Lin + o, slab, slab + p.N, p.Lout + o);
    if (valid)
        for (int64_t t = c.s; t < p.N; t += c.S)                  // decoded_bits: L > 0 in 'decode' mode only (:148-152)
            p.bits[o + t] = (uint8_t)((p.want_bits && p.Lout[o + t] > 0) ? 1 : 0);
}

struct TurboParams {
    MapTables tb;
    const double *sys, *p1, *p2, *Lint;   // [B][N], Lint may be null
    const int32_t *perm;                  // [N]
    uint8_t *bits;                        // [B][N]
    double *scratch;                      // per codeword: A[N] B[N] C[N] pr0[N] beta[(N+1)*S]
    int64_t B, N, slab;
    double nv2;
    int n_iter;
};

__global__ __launch_bounds__(64) void turbo_decode_kernel(TurboParams p) {
    LaneCtx c;
    init_ctx(c, p.tb);
    const int64_t cw = (int64_t)blockIdx.x * c.G + c.g;
    const bool valid = cw < p.B;
    const int64_t N = p.N, o = (valid ? cw : 0) * N;
    double *slab = p.scratch + (valid ? cw : 0) * p.slab;
    double *A = slab, *Bb = slab + N, *C = slab + 2 * N, *pr0 = slab + 3 * N, *beta = slab + 4 * N;
    const int S = c.S;
    if (valid)
        for (int64_t t = c.s; t < N; t += S) A[t] = p.Lint ? p.Lint[o + t] : 0.0;     // L_int_1 (:305-308)
    __syncthreads();
    for (int it = 0; it < p.n_iter; it++) {
        // [L_ext_1, _] = map_decode(sys, non_sys_1, trellis, nv, L_int_1, 'compute')   (:315)
        map_pass(c, valid, N, p.nv2, p.sys + o, nullptr, p.p1 + o, A, pr0, beta, Bb);
        // L_ext_1 -= L_int_1 ; L_int_2 = interlv(L_ext_1)                               (:318-319)
        if (valid)
            for (int64_t t = c.s; t < N; t += S) Bb[t] = Bb[t] - A[t];
        __syncthreads();
        if (valid)
            for (int64_t t = c.s; t < N; t += S) C[t] = Bb[p.perm[t]];
        __syncthreads();
        // [L_2, bits] = map_decode(sys_i, non_sys_2, trellis, nv, L_int_2, mode)          (:326)
        map_pass(c, valid, N, p.nv2, p.sys + o, p.perm, p.p2 + o, C, pr0, beta, Bb);
        // L_ext_2 = L_2 - L_int_2 ; L_int_1 = deinterlv(L_ext_2)                          (:328-329)
        if (valid)
            for (int64_t t = c.s; t < N; t += S) A[p.perm[t]] = Bb[t] - C[t];
        __syncthreads();
    }
    // decoded_bits = deinterlv(decoded_bits of the last MAP2)                              (:331)
    if (valid)
        for (int64_t t = c.s; t < N; t += S)
            p.bits[o + p.perm[t]] = (uint8_t)((p.n_iter > 0 && Bb[t] > 0) ? 1 : 0);
}

int fill_tables(const cpx_trellis *t, MapTables &tb) {
    CPX_REQUIRE(t, CPX_EINVAL, "map_decode: null trellis");
    CPX_REQUIRE(t->I == 2 && t->k == 1, CPX_ELIMIT, "map_decode: only k = 1 (two inputs per step) trellises are supported, like the reference's priors[2]");
    CPX_REQUIRE(t->n >= 2, CPX_EINVAL, "map_decode: needs a rate-1/2 trellis (n >= 2)");
    CPX_REQUIRE(t->S <= 64, CPX_ELIMIT, "map_decode: at most 64 states");
    tb.next_state = t->d_next; tb.output = t->d_out;
    tb.pred_state = t->d_pred_state; tb.pred_input = t->d_pred_input; tb.pred_code = t->d_pred_code;
    tb.n = t->n;
    tb.lgS = 0;
    while ((1 << tb.lgS) < t->S) tb.lgS++;
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
    const int S = t->S, G = 64 / S;
    p.sys = d_sys; p.par = d_par; p.Lin = d_L_int; p.Lout = d_L_ext; p.bits = d_bits;
    p.B = B; p.N = N; p.nv2 = 2 * noise_variance; p.want_bits = want_bits;
    p.slab = N + (N + 1) * S;
    if ((rc = workspace(st, 0, sizeof(double) * (size_t)(p.slab * B), (void **)&p.scratch))) return rc;
    hipLaunchKernelGGL(map_decode_kernel, dim3((unsigned)((B + G - 1) / G)), dim3(64), 0, st, p);
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
    const int S = t->S, G = 64 / S;
    p.sys = d_sys; p.p1 = d_p1; p.p2 = d_p2; p.Lint = d_L_int_or_null; p.perm = d_perm; p.bits = d_bits;
    p.B = B; p.N = N; p.nv2 = 2 * noise_variance; p.n_iter = n_iter;
    p.slab = 4 * N + (N + 1) * S;
    if ((rc = workspace(st, 0, sizeof(double) * (size_t)(p.slab * B), (void **)&p.scratch))) return rc;
    hipLaunchKernelGGL(turbo_decode_kernel, dim3((unsigned)((B + G - 1) / G)), dim3(64), 0, st, p);
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
