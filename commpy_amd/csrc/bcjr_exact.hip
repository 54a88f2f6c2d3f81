// Absolute-scale BCJR / turbo decoder: the redo path behind bcjr.hip's "detect and redo", and the ONLY path for trellises of
// more than 16 states (the reference's map_decode takes any number of states; flags == null decodes every codeword).
//
// A literal restatement, one codeword per lane, of
//   map_decode    (/root/reference/commpy/channelcoding/turbo.py:163-251)
//     _compute_branch_prob (:62-76), _backward_recursion (:78-111), _forward_recursion_decoding (:114-158)
//   turbo_decode  (turbo.py:254-333) with interlv / deinterlv (interleavers.py:13-47)
// in the reference's own scale and evaluation order: absolute gamma = exp(-(x^2 + y^2) / (2 sigma^2)), priors
// p0 = 1 / (1 + e^L), p1 = 1 - p0, every beta / alpha column divided by its NumPy-ordered sum at every step.  Where the
// terms of that recursion underflow, the reference returns NaN or +-inf LLRs (a column sum of 0, app0 = 0); bcjr.hip's fast,
// scale-free kernels would return finite values there, so they flag every codeword for which any term of the reference can
// leave the normal float64 range and the kernels below decode exactly those codewords again, overwriting the outputs.
// Speed is not a goal: lane = codeword, the backward metrics of a lane live in an HBM scratch laid out [t][state][lane]
// (coalesced over the lanes).  With no flag raised a launch is C / 64 wavefronts that read B / C flag bytes per lane and exit.
#include "cpx_internal.h"

#include <algorithm>

using namespace cpx;

namespace {

constexpr int MAXS = 16;        // bcjr.hip's fast kernels support 2..16 states: their state vectors fit a lane's registers here.
                                // Larger trellises (the reference takes any, turbo.py:163-251) keep them in the HBM scratch (MemVec)

struct ExCode {
    const int32_t *next_state, *output;   // [S][2]
    int S, n;
};

// strided array of one lane; `perm` (may be null) gathers: element t is p[perm[t] * st]  (interlv: out = in[p])
struct View {
    const double *p;
    int64_t st;
    const int32_t *perm;
    __device__ __forceinline__ double operator()(int64_t t) const { return p[(perm ? (int64_t)perm[t] : t) * st]; }
};

// NumPy float64 add.reduce over n <= 128 values a(0 .. n-1) (loops_utils.h.src DOUBLE_pairwise_sum): n < 8 sequential from 0,
// else eight accumulators combined pairwise, remainder added sequentially -- turbo.py:110-111, :155-156 rely on it
template <class A>
__device__ __forceinline__ double np_sum_leaf(const A &a, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; i++) res += a(i);
        return res;
    }
    double r[8];
    for (int j = 0; j < 8; j++) r[j] = a(j);
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; j++) r[j] += a(i + j);
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a(i);
    return res;
}
// ... and above 128 values: split at n/2 rounded down to a multiple of 8, left half first (the recursion of pairwise_sum as
// an explicit stack: 65536 states are ten levels)
template <class A>
__device__ double np_sum(const A &a, int n) {
    if (n <= 128) return np_sum_leaf(a, n);
    struct Frame { int off, n, stage; double left; } stk[12];
    int sp = 0;
    stk[0] = Frame{0, n, 0, 0.0};
    double ret = 0.0;
    while (sp >= 0) {
        Frame &f = stk[sp];
        int n2 = f.n / 2;
        n2 -= n2 % 8;
        if (f.stage == 0) {
            if (f.n <= 128) {
                const int off = f.off;
                ret = np_sum_leaf([&](int i) { return a(off + i); }, f.n);
                sp--;
            } else {
                f.stage = 1;
                stk[++sp] = Frame{f.off, n2, 0, 0.0};
            }
        } else if (f.stage == 1) {
            f.left = ret;
            f.stage = 2;
            stk[++sp] = Frame{f.off + n2, f.n - n2, 0, 0.0};
        } else {
            ret = f.left + ret;
            sp--;
        }
    }
    return ret;
}

// the three state vectors of a pass (sums of a column before normalisation, alpha, next alpha): a lane's registers up to
// MAXS states, else three rows [S] of the lane's HBM scratch (lane stride C)
struct RegVec {
    double acc_[MAXS], f_[MAXS], fn_[MAXS];
    __device__ __forceinline__ double &acc(int s) { return acc_[s]; }
    __device__ __forceinline__ double &f(int s) { return f_[s]; }
    __device__ __forceinline__ double &fn(int s) { return fn_[s]; }
};
struct MemVec {
    double *base;              // [3][S] with lane stride C
    int64_t C;
    int S;
    __device__ __forceinline__ double &acc(int s) { return base[(int64_t)s * C]; }
    __device__ __forceinline__ double &f(int s) { return base[((int64_t)S + s) * C]; }
    __device__ __forceinline__ double &fn(int s) { return base[((int64_t)2 * S + s) * C]; }
};

// One MAP pass of one codeword (turbo.py:163-251).  beta: scratch [(N + 1)][S] with lane stride C; Lout: stride lst;
// bits (may be null): stride 1, written at position (bperm ? bperm[t] : t) -- deinterlv of turbo_decode's last pass (:331).
template <class Vec>
__device__ void exact_map(const ExCode &cd, int64_t N, double nv2, const View &sys, const View &par, const View &lin,
                          double *beta, int64_t C, double *Lout, int64_t lst, uint8_t *bits, const int32_t *bperm, int want_bits,
                          Vec &v) {
    const int S = cd.S, sh = cd.n - 2;
    auto B_ = [&](int64_t t, int s) -> double & { return beta[(t * S + s) * C]; };
    auto gammas = [&](int64_t t, double (&g)[4]) {                // _compute_branch_prob for the four (msg_bit, parity_bit) pairs
        const double r0 = sys(t), r1 = par(t);
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const double x = r0 - (double)(2 * (c >> 1) - 1);     // code_symbol = 2 * code_bit - 1 (:67-71)
            const double y = r1 - (double)(2 * (c & 1) - 1);
            g[c] = exp(-(x * x + y * y) / nv2);                   // (:74); nv2 = 2 * noise_variance
        }
    };
    for (int s = 0; s < S; s++) B_(N, s) = 1.0;                   // b_state_metrics[:, N] = 1 (:225)
    for (int64_t rt = N; rt >= 1; --rt) {                         // _backward_recursion (:93-111)
        double g[4];
        gammas(rt - 1, g);
        const double p0 = 1.0 / (1.0 + exp(lin(rt - 1)));         // priors (:239-240)
        const double pr[2] = {p0, 1.0 - p0};
        for (int cs = 0; cs < S; cs++) {
            double a = 0.0;
            for (int ci = 0; ci < 2; ci++) {
                const int ns = cd.next_state[cs * 2 + ci], code = (cd.output[cs * 2 + ci] >> sh) & 3;   // [msg_bit, parity_bit] (:96-98)
                a += (B_(rt, ns) * g[code] * pr[ci]);             // (:106-108)
            }
            v.acc(cs) = a;
        }
        const double sum = np_sum([&](int s) { return v.acc(s); }, S);   // (:110-111)
        for (int s = 0; s < S; s++) B_(rt - 1, s) = v.acc(s) / sum;
    }
    for (int s = 0; s < S; s++) { v.f(s) = (s == 0) ? 1.0 : 0.0; v.fn(s) = 0.0; }   // f_state_metrics[0][0] = 1 (:221)
    for (int64_t t = 1; t <= N; t++) {                            // _forward_recursion_decoding (:127-158)
        double g[4];
        gammas(t - 1, g);
        const double li = lin(t - 1);
        const double p0 = 1.0 / (1.0 + exp(li));
        const double pr[2] = {p0, 1.0 - p0};
        double app[2] = {0.0, 0.0};
        for (int cs = 0; cs < S; cs++)
            for (int ci = 0; ci < 2; ci++) {
                const int ns = cd.next_state[cs * 2 + ci], code = (cd.output[cs * 2 + ci] >> sh) & 3;
                v.fn(ns) += (v.f(cs) * g[code] * pr[ci]);         // (:136-138)
                app[ci] += (v.f(cs) * g[code] * B_(t, ns));       // (:141-143)
            }
        const double lappr = li + log(app[1] / app[0]);           // (:145)
        Lout[(t - 1) * lst] = lappr;
        if (bits) bits[bperm ? bperm[t - 1] : t - 1] = (uint8_t)((want_bits && lappr > 0) ? 1 : 0);   // (:148-152)
        const double sum = np_sum([&](int s) { return v.fn(s); }, S);   // (:155)
        for (int s = 0; s < S; s++) { v.f(s) = v.fn(s) / sum; v.fn(s) = 0.0; }
    }
}

struct ExMapParams {
    ExCode cd;
    const double *sys, *par, *Lin;   // [B][N]
    double *Lout;                    // [B][N]
    uint8_t *bits;                   // [B][N]
    const uint8_t *flags;            // [B], or null: every codeword (trellises the fast kernels do not serve)
    double *scratch;                 // beta: [(N + 1) * S][C] (+ [3 S][C] state vectors when S > MAXS)
    int64_t B, N, C;
    double nv2;
    int want_bits;
};

// the state vectors of lane j: registers, or three rows behind the `used` doubles of its scratch column
template <bool BIG> struct VecOf;
template <> struct VecOf<false> {
    RegVec v;
    __device__ VecOf(double *, int64_t, int64_t, int) {}
};
template <> struct VecOf<true> {
    MemVec v;
    __device__ VecOf(double *col, int64_t used, int64_t C, int S) : v{col + used * C, C, S} {}
};

template <bool BIG>
__global__ __launch_bounds__(64) void map_exact_kernel(ExMapParams p) {
    const int64_t j = (int64_t)blockIdx.x * 64 + threadIdx.x;    // lane = scratch column; decodes the flagged codewords = j (mod C)
    VecOf<BIG> vec(p.scratch + j, (p.N + 1) * p.cd.S, p.C, p.cd.S);
    for (int64_t cw = j; cw < p.B; cw += p.C) {
        if (p.flags && !p.flags[cw]) continue;
        const int64_t o = cw * p.N;
        const View sys{p.sys + o, 1, nullptr}, par{p.par + o, 1, nullptr}, lin{p.Lin + o, 1, nullptr};
        exact_map(p.cd, p.N, p.nv2, sys, par, lin, p.scratch + j, p.C, p.Lout + o, 1, p.bits ? p.bits + o : nullptr, nullptr,
                  p.want_bits, vec.v);
    }
}

struct ExTurboParams {
    ExCode cd;
    const double *sys, *p1, *p2, *Lint;   // [B][N], Lint may be null
    const int32_t *perm;                  // [N]
    uint8_t *bits;                        // [B][N]
    const uint8_t *flags;                 // [B], or null: every codeword
    double *scratch;                      // [(N + 1) * S + 4 N][C]: beta, L_int_1, L_ext, L_int_2, L_2 (+ [3 S][C] when S > MAXS)
    int64_t B, N, C;
    double nv2;
    int n_iter;
};

template <bool BIG>
__global__ __launch_bounds__(64) void turbo_exact_kernel(ExTurboParams p) {
    const int64_t j = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t N = p.N, C = p.C;
    double *beta = p.scratch + j;
    double *L1 = beta + (N + 1) * p.cd.S * C, *Le = L1 + N * C, *L2in = Le + N * C, *L2 = L2in + N * C;
    VecOf<BIG> vec(p.scratch + j, (N + 1) * p.cd.S + 4 * N, C, p.cd.S);
    for (int64_t cw = j; cw < p.B; cw += C) {
        if (p.flags && !p.flags[cw]) continue;
        const int64_t o = cw * N;
        uint8_t *bits = p.bits + o;
        for (int64_t i = 0; i < N; i++) {
            L1[i * C] = p.Lint ? p.Lint[o + i] : 0.0;             // L_int_1 (:305-308)
            bits[i] = 0;                                          // number_iterations = 0: zeros (:302, :331)
        }
        const View sys{p.sys + o, 1, nullptr}, sysi{p.sys + o, 1, p.perm};          // interlv(sys_symbols) (:310)
        const View y1{p.p1 + o, 1, nullptr}, y2{p.p2 + o, 1, nullptr};
        for (int it = 0; it < p.n_iter; it++) {
            exact_map(p.cd, N, p.nv2, sys, y1, View{L1, C, nullptr}, beta, C, Le, C, nullptr, nullptr, 0, vec.v);   // 'compute' (:315)
            for (int64_t i = 0; i < N; i++) Le[i * C] = Le[i * C] - L1[i * C];                           // (:318)
            for (int64_t i = 0; i < N; i++) L2in[i * C] = Le[(int64_t)p.perm[i] * C];                    // interlv (:319)
            const bool last = it == p.n_iter - 1;                 // mode 'decode' in the last iteration only (:320-323)
            exact_map(p.cd, N, p.nv2, sysi, y2, View{L2in, C, nullptr}, beta, C, L2, C, last ? bits : nullptr, p.perm, 1, vec.v);   // (:326, :331)
            for (int64_t i = 0; i < N; i++) L1[(int64_t)p.perm[i] * C] = L2[i * C] - L2in[i * C];       // deinterlv (:328-329)
        }
    }
}

// lanes of the redo launch: a multiple of 64, at most one per codeword, at most `budget` bytes of scratch
int64_t pick_lanes(int64_t B, size_t doubles_per_lane, size_t budget) {
    int64_t c = (int64_t)(budget / (doubles_per_lane * sizeof(double)));
    c = std::min<int64_t>(c, 4096);
    c = std::min<int64_t>(c, (B + 63) / 64 * 64);
    return c / 64 * 64;
}

ExCode ex_code(const cpx_trellis *t) { return ExCode{t->d_next, t->d_out, t->S, t->n}; }

constexpr size_t EXACT_BUDGET = (size_t)1 << 28;     // 256 MB: 4096 lanes for N = 1024, four states
constexpr size_t EXACT_BUDGET_BIG = (size_t)1 << 32; // trellises only this path serves: 4 GB (64 lanes of 1024 steps x 8192 states)

}  // namespace

namespace cpx {

static size_t lane_doubles(int S, int64_t N, int turbo) {
    return (size_t)((N + 1) * S + (turbo ? 4 * N : 0) + (S > MAXS ? 3 * S : 0));
}

bool bcjr_exact_supported(int S, int64_t N, int turbo) {
    return pick_lanes(64, lane_doubles(S, N, turbo), S > MAXS ? EXACT_BUDGET_BIG : EXACT_BUDGET) >= 64;
}

int bcjr_exact_map(const cpx_trellis *t, const double *sys, const double *par, const double *Lin, int64_t B, int64_t N, double nv2,
                   int want_bits, double *Lout, uint8_t *bits, const uint8_t *flags, hipStream_t st) {
    ExMapParams p;
    p.cd = ex_code(t);
    const size_t per = lane_doubles(t->S, N, 0);
    p.C = pick_lanes(B, per, t->S > MAXS ? EXACT_BUDGET_BIG : EXACT_BUDGET);
    CPX_REQUIRE(p.C >= 64, CPX_ELIMIT, "map_decode: block too long for the absolute-scale path (%d states x %lld steps)", t->S, (long long)N);
    void *sc = nullptr;
    if (int rc = workspace(st, 5, per * (size_t)p.C * sizeof(double), &sc)) return rc;
    p.sys = sys; p.par = par; p.Lin = Lin; p.Lout = Lout; p.bits = bits; p.flags = flags; p.scratch = static_cast<double *>(sc);
    p.B = B; p.N = N; p.nv2 = nv2; p.want_bits = want_bits;
    if (t->S > MAXS) hipLaunchKernelGGL(map_exact_kernel<true>, dim3((unsigned)(p.C / 64)), dim3(64), 0, st, p);
    else hipLaunchKernelGGL(map_exact_kernel<false>, dim3((unsigned)(p.C / 64)), dim3(64), 0, st, p);
    CPX_HIP(hipGetLastError());
    return CPX_OK;
}

int bcjr_exact_turbo(const cpx_trellis *t, const double *sys, const double *p1, const double *p2, const double *Lint_or_null,
                     const int32_t *perm, int64_t B, int64_t N, double nv2, int n_iter, uint8_t *bits, const uint8_t *flags,
                     hipStream_t st) {
    ExTurboParams p;
    p.cd = ex_code(t);
    const size_t per = lane_doubles(t->S, N, 1);
    p.C = pick_lanes(B, per, t->S > MAXS ? EXACT_BUDGET_BIG : EXACT_BUDGET);
    CPX_REQUIRE(p.C >= 64, CPX_ELIMIT, "turbo_decode: block too long for the absolute-scale path (%d states x %lld steps)", t->S, (long long)N);
    void *sc = nullptr;
    if (int rc = workspace(st, 5, per * (size_t)p.C * sizeof(double), &sc)) return rc;
    p.sys = sys; p.p1 = p1; p.p2 = p2; p.Lint = Lint_or_null; p.perm = perm; p.bits = bits; p.flags = flags;
    p.scratch = static_cast<double *>(sc);
    p.B = B; p.N = N; p.nv2 = nv2; p.n_iter = n_iter;
    if (t->S > MAXS) hipLaunchKernelGGL(turbo_exact_kernel<true>, dim3((unsigned)(p.C / 64)), dim3(64), 0, st, p);
    else hipLaunchKernelGGL(turbo_exact_kernel<false>, dim3((unsigned)(p.C / 64)), dim3(64), 0, st, p);
    CPX_HIP(hipGetLastError());
    return CPX_OK;
}

}  // namespace cpx
