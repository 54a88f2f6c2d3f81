// Absolute-scale BCJR / turbo decoder: the redo path behind bcjr.hip's "detect and redo".
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

constexpr int MAXS = 16;        // bcjr.hip supports 2..16 states

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

// NumPy float64 add.reduce over n <= 16 values (loops_utils.h.src DOUBLE_pairwise_sum): n < 8 sequential from 0, else eight
// accumulators combined pairwise, remainder added sequentially -- turbo.py:110-111, :155-156 rely on it
__device__ __forceinline__ double np_sum16(const double *a, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    }
    double r[8];
    for (int j = 0; j < 8; j++) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; j++) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

// One MAP pass of one codeword (turbo.py:163-251).  beta: scratch [(N + 1)][S] with lane stride C; Lout: stride lst;
// bits (may be null): stride 1, written at position (bperm ? bperm[t] : t) -- deinterlv of turbo_decode's last pass (:331).
__device__ void exact_map(const ExCode &cd, int64_t N, double nv2, const View &sys, const View &par, const View &lin,
                          double *beta, int64_t C, double *Lout, int64_t lst, uint8_t *bits, const int32_t *bperm, int want_bits) {
    const int S = cd.S, sh = cd.n - 2;
    double acc[MAXS], f[MAXS], fn[MAXS];
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
            acc[cs] = a;
        }
        const double sum = np_sum16(acc, S);                      // (:110-111)
        for (int s = 0; s < S; s++) B_(rt - 1, s) = acc[s] / sum;
    }
    for (int s = 0; s < S; s++) { f[s] = (s == 0) ? 1.0 : 0.0; fn[s] = 0.0; }   // f_state_metrics[0][0] = 1 (:221)
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
                fn[ns] += (f[cs] * g[code] * pr[ci]);             // (:136-138)
                app[ci] += (f[cs] * g[code] * B_(t, ns));         // (:141-143)
            }
        const double lappr = li + log(app[1] / app[0]);           // (:145)
        Lout[(t - 1) * lst] = lappr;
        if (bits) bits[bperm ? bperm[t - 1] : t - 1] = (uint8_t)((want_bits && lappr > 0) ? 1 : 0);   // (:148-152)
        const double sum = np_sum16(fn, S);                       // (:155)
        for (int s = 0; s < S; s++) { f[s] = fn[s] / sum; fn[s] = 0.0; }
    }
}

struct ExMapParams {
    ExCode cd;
    const double *sys, *par, *Lin;   // [B][N]
    double *Lout;                    // [B][N]
    uint8_t *bits;                   // [B][N]
    const uint8_t *flags;            // [B]
    double *scratch;                 // beta: [(N + 1) * S][C]
    int64_t B, N, C;
    double nv2;
    int want_bits;
};

__global__ __launch_bounds__(64) void map_exact_kernel(ExMapParams p) {
    const int64_t j = (int64_t)blockIdx.x * 64 + threadIdx.x;    // lane = scratch column; decodes the flagged codewords = j (mod C)
    for (int64_t cw = j; cw < p.B; cw += p.C) {
        if (!p.flags[cw]) continue;
        const int64_t o = cw * p.N;
        const View sys{p.sys + o, 1, nullptr}, par{p.par + o, 1, nullptr}, lin{p.Lin + o, 1, nullptr};
        exact_map(p.cd, p.N, p.nv2, sys, par, lin, p.scratch + j, p.C, p.Lout + o, 1, p.bits ? p.bits + o : nullptr, nullptr,
                  p.want_bits);
    }
}

struct ExTurboParams {
    ExCode cd;
    const double *sys, *p1, *p2, *Lint;   // [B][N], Lint may be null
    const int32_t *perm;                  // [N]
    uint8_t *bits;                        // [B][N]
    const uint8_t *flags;                 // [B]
    double *scratch;                      // [(N + 1) * S + 4 N][C]: beta, L_int_1, L_ext, L_int_2, L_2
    int64_t B, N, C;
    double nv2;
    int n_iter;
};

__global__ __launch_bounds__(64) void turbo_exact_kernel(ExTurboParams p) {
    const int64_t j = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t N = p.N, C = p.C;
    double *beta = p.scratch + j;
    double *L1 = beta + (N + 1) * p.cd.S * C, *Le = L1 + N * C, *L2in = Le + N * C, *L2 = L2in + N * C;
    for (int64_t cw = j; cw < p.B; cw += C) {
        if (!p.flags[cw]) continue;
        const int64_t o = cw * N;
        uint8_t *bits = p.bits + o;
        for (int64_t i = 0; i < N; i++) {
            L1[i * C] = p.Lint ? p.Lint[o + i] : 0.0;             // L_int_1 (:305-308)
            bits[i] = 0;                                          // number_iterations = 0: zeros (:302, :331)
        }
        const View sys{p.sys + o, 1, nullptr}, sysi{p.sys + o, 1, p.perm};          // interlv(sys_symbols) (:310)
        const View y1{p.p1 + o, 1, nullptr}, y2{p.p2 + o, 1, nullptr};
        for (int it = 0; it < p.n_iter; it++) {
            exact_map(p.cd, N, p.nv2, sys, y1, View{L1, C, nullptr}, beta, C, Le, C, nullptr, nullptr, 0);   // 'compute' (:315)
            for (int64_t i = 0; i < N; i++) Le[i * C] = Le[i * C] - L1[i * C];                           // (:318)
            for (int64_t i = 0; i < N; i++) L2in[i * C] = Le[(int64_t)p.perm[i] * C];                    // interlv (:319)
            const bool last = it == p.n_iter - 1;                 // mode 'decode' in the last iteration only (:320-323)
            exact_map(p.cd, N, p.nv2, sysi, y2, View{L2in, C, nullptr}, beta, C, L2, C, last ? bits : nullptr, p.perm, 1);   // (:326, :331)
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

}  // namespace

namespace cpx {

bool bcjr_exact_supported(int S, int64_t N, int turbo) {
    return pick_lanes(64, (size_t)((N + 1) * S + (turbo ? 4 * N : 0)), EXACT_BUDGET) >= 64;
}

int bcjr_exact_map(const cpx_trellis *t, const double *sys, const double *par, const double *Lin, int64_t B, int64_t N, double nv2,
                   int want_bits, double *Lout, uint8_t *bits, const uint8_t *flags, hipStream_t st) {
    ExMapParams p;
    p.cd = ex_code(t);
    const size_t per = (size_t)((N + 1) * t->S);
    p.C = pick_lanes(B, per, EXACT_BUDGET);
    CPX_REQUIRE(p.C >= 64, CPX_ELIMIT, "map_decode: block too long for the absolute-scale redo path");
    void *sc = nullptr;
    if (int rc = workspace(st, 5, per * (size_t)p.C * sizeof(double), &sc)) return rc;
    p.sys = sys; p.par = par; p.Lin = Lin; p.Lout = Lout; p.bits = bits; p.flags = flags; p.scratch = static_cast<double *>(sc);
    p.B = B; p.N = N; p.nv2 = nv2; p.want_bits = want_bits;
    hipLaunchKernelGGL(map_exact_kernel, dim3((unsigned)(p.C / 64)), dim3(64), 0, st, p);
    CPX_HIP(hipGetLastError());
    return CPX_OK;
}

int bcjr_exact_turbo(const cpx_trellis *t, const double *sys, const double *p1, const double *p2, const double *Lint_or_null,
                     const int32_t *perm, int64_t B, int64_t N, double nv2, int n_iter, uint8_t *bits, const uint8_t *flags,
                     hipStream_t st) {
    ExTurboParams p;
    p.cd = ex_code(t);
    const size_t per = (size_t)((N + 1) * t->S + 4 * N);
    p.C = pick_lanes(B, per, EXACT_BUDGET);
    CPX_REQUIRE(p.C >= 64, CPX_ELIMIT, "turbo_decode: block too long for the absolute-scale redo path");
    void *sc = nullptr;
    if (int rc = workspace(st, 5, per * (size_t)p.C * sizeof(double), &sc)) return rc;
    p.sys = sys; p.p1 = p1; p.p2 = p2; p.Lint = Lint_or_null; p.perm = perm; p.bits = bits; p.flags = flags;
    p.scratch = static_cast<double *>(sc);
    p.B = B; p.N = N; p.nv2 = nv2; p.n_iter = n_iter;
    hipLaunchKernelGGL(turbo_exact_kernel, dim3((unsigned)(p.C / 64)), dim3(64), 0, st, p);
    CPX_HIP(hipGetLastError());
    return CPX_OK;
}

}  // namespace cpx
