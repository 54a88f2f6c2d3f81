// PSK/QAM demodulation for gfx950.  Replaces the body of Modem.demodulate
// (/root/reference/commpy/modulation.py:100-141):
//   soft (:125-137): LLR(b) = log( sum_{m:(m>>b)&1} e^{-|y-c_m|^2/noise_var} / sum_{m:!(..)} ... ),
//                    written at index i*nb + nb-1-b, sums in increasing constellation index m;
//   hard (:121-123): first-minimum nearest point |y-c_m| -> MSB-first nb bits (int8).
// Element-wise map: one received symbol per lane, tables (<= 256 points, 4 KiB) staged in LDS and read
// as wave-uniform broadcasts (larger constellations, up to 65536 points: the same formulas with the table read from HBM).  float64 like the reference.  Each symbol is read once (16 B) and nb
// outputs written once: HBM traffic equals the algorithmic bytes (16 + 8*nb per symbol).
//
// Two code paths, chosen when the modem handle is created:
//   * generic constellation (PSK, custom): the reference's naive formula, M exps per symbol, |.| is
//     hypot like np.abs of a complex scalar;
//   * axis-separable square QAM (every constellation QAMModem builds): label m = (a << NH) | b with
//     c_m = xs[a] + 1j*ys[b], so e_m = ex[a]*ey[b] and a bit of `a` only needs the real axis:
//       sum_{m:bit} e_m / sum_{m:!bit} e_m = (sum_{a:bit} ex[a] * sum_b ey[b]) / (sum_{a:!bit} ex[a] * sum_b ey[b])
//     i.e. 2*sqrt(M) exps per symbol instead of M (64-QAM: 16 vs 64).  Numerator and denominator are
//     still formed as those products so that the reference's underflow pattern at very high SNR
//     (log(0/x) = -inf, 0/0 = NaN, modulation.py:134-137) is kept; finite values differ from the
//     sequential sums by O(1e-16) relative (tolerance 1e-5).  Hard decisions decompose per axis
//     exactly (nearest grid point, first minimum = lowest label); symbols whose two best axis
//     distances are closer than 1e-12 relative are re-decided with the reference's full hypot scan.
#include "cpx_internal.h"
#include "cpx_math.h"
#include "demod_dev.h"
#include "cpx_rng.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>

using namespace cpx;

namespace {

constexpr int DEMOD_BLOCK = 256;
constexpr int MAX_M = 256;

// `scale` multiplies every LLR on its way out (1.0: exact identity; -1.0: the sign flip between Modem.demodulate -- log P1/P0 --
// and ldpc_bp_decode -- log P0/P1 --, test_ldpc.py:53-54, without a second pass over the LLRs).
template <int NB, bool RCP>
__global__ __launch_bounds__(DEMOD_BLOCK) void demod_soft_kernel(const double2 *__restrict__ y, int64_t Ns,
                                                                 const double2 *__restrict__ cst, int M,
                                                                 double noise_var, double scale, double *__restrict__ llr) {
    __shared__ double2 c_s[MAX_M];
    for (int m = threadIdx.x; m < M; m += DEMOD_BLOCK) c_s[m] = cst[m];
    __syncthreads();
    const double ninv = -1.0 / noise_var;                             // RCP: see demod_soft_sep_kernel
    for (int64_t i = (int64_t)blockIdx.x * DEMOD_BLOCK + threadIdx.x; i < Ns; i += (int64_t)gridDim.x * DEMOD_BLOCK) {
        const double2 cur = y[i];
        double num[NB], den[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) { num[b] = 0.0; den[b] = 0.0; }
        for (int m = 0; m < M; m++) {
            const double2 c = c_s[m];
            const double a = hypot(cur.x - c.x, cur.y - c.y);       // abs(current_symbol - symbol)
            const double e = exp(RCP ? (a * a) * ninv : (-(a * a)) / noise_var);   // exp((-abs(..)**2)/noise_var) (:134,136)
#pragma unroll
            for (int b = 0; b < NB; b++) {
                if ((m >> b) & 1) num[b] += e; else den[b] += e;
            }
        }
#pragma unroll
        for (int b = 0; b < NB; b++) llr[i * NB + NB - 1 - b] = fast_log(num[b] / den[b]) * scale;   // (:137)
    }
}

// RCP: the exponent -d^2 / noise_var is formed as d^2 * (-1 / noise_var) -- one multiplication instead of a 12-instruction
// division per exponential (16 of them per 64-QAM symbol, a sixth of the kernel); the two agree to an ulp of the exponent,
// i.e. ~1e-14 on an LLR (the bar is 1e-5).  The host selects it when 1 / noise_var is a normal number.
// GP (round 4): the R levels of an axis are EQUALLY SPACED and labelled in reflected Gray order -- every constellation QAMModem
// builds (modulation.py:258-261 + the Gray re-indexing of :72-75): level j = p0 + j d carries label j ^ (j >> 1).  Then the
// exponentials of an axis form a geometric progression of geometric progressions,
//     e[j+1] = e[j] rho[j],  rho[j] = exp((2 d (x - p[j]) - d^2) / N0),  rho[j+1] = rho[j] Q,  Q = exp(-2 d^2 / N0)
// (and the mirror image downwards), so FOUR exp per axis -- the two middle levels and their two first ratios -- give all R values,
// the rest by two multiplications per level: 64-QAM 8 instead of 16 exp per symbol, 256-QAM 8 instead of 32 (round 5: TWO exp and a
// division per axis, see axis_gp).  Starting in the
// middle keeps every factor that matters a normal number: the results carry a few more roundings than exp itself (~1e-15 on an
// LLR).  Where they could not -- a middle-level exponential below 1e-290, an axis whose sum is below 1e-30, or anything non-finite --
// the symbol joins the point-by-point redo below (which the +-600 rule already sends the deep-underflow cases to); with both
// guards passed and every |LLR| < 600, every term that contributes to a sum at the 1e-7 level is above 1e-300.
// ---- table-driven exp / log for the separable kernel (round 5) -----------------------------------------------------------------------
// The geometric-progression kernel is VALU bound on four `exp` (37 instructions each) and six `log` (36 each, of a quotient that
// took nine more) per 64-QAM symbol.  With 32-entry tables in LDS -- 2^(j/32); 1/c_i and -log(1/c_i) for the 32 intervals of a
// mantissa in [0.5, 1) -- an exponential is 16 instructions (x = (32 k' + j) ln2/32 + r, |r| <= ln2/64: 2^k' T[j] (1 + r + .. + r^6/720))
// and a logarithm 17 (m invc[i] - 1 = r, |r| <= 1/64: k ln2 + logc[i] + r - r^2/2 + .. - r^8/8), both to an ulp or two of the result
// (the bar on an LLR is 1e-5; measured against the oracle: tests/test_bcjr_ldpc_demod_gpu.py, bench.py other_configs).  A 32-entry
// table of doubles covers the 64 LDS banks exactly once: lanes with equal indices are a broadcast, others never collide.  Special
// values: a NaN comes out as a NaN, an argument past the range as 0 / inf / a denormal through ldexp -- every such symbol fails the
// kernel's own guards and is decided point by point with the library functions, as before.
__device__ const double DEMOD_TAB[96] = {
    // 2^(j/32)
    0x1.0000000000000p+0, 0x1.059b0d3158574p+0, 0x1.0b5586cf9890fp+0, 0x1.11301d0125b51p+0,
    0x1.172b83c7d517bp+0, 0x1.1d4873168b9aap+0, 0x1.2387a6e756238p+0, 0x1.29e9df51fdee1p+0,
    0x1.306fe0a31b715p+0, 0x1.371a7373aa9cbp+0, 0x1.3dea64c123422p+0, 0x1.44e086061892dp+0,
    0x1.4bfdad5362a27p+0, 0x1.5342b569d4f82p+0, 0x1.5ab07dd485429p+0, 0x1.6247eb03a5585p+0,
    0x1.6a09e667f3bcdp+0, 0x1.71f75e8ec5f74p+0, 0x1.7a11473eb0187p+0, 0x1.82589994cce13p+0,
    0x1.8ace5422aa0dbp+0, 0x1.93737b0cdc5e5p+0, 0x1.9c49182a3f090p+0, 0x1.a5503b23e255dp+0,
    0x1.ae89f995ad3adp+0, 0x1.b7f76f2fb5e47p+0, 0x1.c199bdd85529cp+0, 0x1.cb720dcef9069p+0,
    0x1.d5818dcfba487p+0, 0x1.dfc97337b9b5fp+0, 0x1.ea4afa2a490dap+0, 0x1.f50765b6e4540p+0,
    // invc[i] = double(1 / (0.5 + (i + 0.5) / 64))
    0x1.f81f81f81f820p+0, 0x1.e9131abf0b767p+0, 0x1.dae6076b981dbp+0, 0x1.cd85689039b0bp+0,
    0x1.c0e070381c0e0p+0, 0x1.b4e81b4e81b4fp+0, 0x1.a98ef606a63bep+0, 0x1.9ec8e951033d9p+0,
    0x1.948b0fcd6e9e0p+0, 0x1.8acb90f6bf3aap+0, 0x1.8181818181818p+0, 0x1.78a4c8178a4c8p+0,
    0x1.702e05c0b8170p+0, 0x1.6816816816817p+0, 0x1.6058160581606p+0, 0x1.58ed2308158edp+0,
    0x1.51d07eae2f815p+0, 0x1.4afd6a052bf5bp+0, 0x1.446f86562d9fbp+0, 0x1.3e22cbce4a902p+0,
    0x1.3813813813814p+0, 0x1.323e34a2b10bfp+0, 0x1.2c9fb4d812ca0p+0, 0x1.27350b8812735p+0,
    0x1.21fb78121fb78p+0, 0x1.1cf06ada2811dp+0, 0x1.1811811811812p+0, 0x1.135c81135c811p+0,
    0x1.0ecf56be69c90p+0, 0x1.0a6810a6810a7p+0, 0x1.0624dd2f1a9fcp+0, 0x1.0204081020408p+0,
    // logc[i] = -log(invc[i])
    -0x1.5af405c3649e0p-1, -0x1.4b6fd6f970c1fp-1, -0x1.3c6080c36bfb5p-1, -0x1.2dbf557b0df43p-1,
    -0x1.1f8635fc61658p-1, -0x1.11af823c75aa8p-1, -0x1.04360be7603aep-1, -0x1.ee2a156b413e5p-2,
    -0x1.d490246defa6ap-2, -0x1.bb9611b80e2fcp-2, -0x1.a33440224fa79p-2, -0x1.8b639a88b2df4p-2,
    -0x1.741d876c67bb1p-2, -0x1.5d5bddf595f31p-2, -0x1.4718dc271c41cp-2, -0x1.314f1e1d35ce3p-2,
    -0x1.1bf99635a6b95p-2, -0x1.07138604d5864p-2, -0x1.e530effe71013p-3, -0x1.bd087383bd8aap-3,
    -0x1.95a5adcf70182p-3, -0x1.6f0128b756ab9p-3, -0x1.4913d8333b563p-3, -0x1.23d712a49c201p-3,
    -0x1.fe89139dbd565p-4, -0x1.b6ac88dad5b1dp-4, -0x1.700d30aeac0e8p-4, -0x1.2aa04a44717a1p-4,
    -0x1.ccb73cdddb2d0p-5, -0x1.466aed42de3f9p-5, -0x1.8492528c8cac5p-6, -0x1.010157588de69p-7
};

__device__ __forceinline__ double tab_exp(double x, const double *__restrict__ tab) {
    constexpr double INV_L32 = 0x1.71547652b82fep+5, L32_HI = 0x1.62e42fefa39efp-6, L32_LO = 0x1.ac00000000000p-61;
    const double k = __builtin_rint(x * INV_L32);
    double r = __builtin_fma(-k, L32_HI, x);
    r = __builtin_fma(-k, L32_LO, r);
    const int ki = (int)k;                                        // v_cvt_i32_f64 saturates; a NaN gives 0 (r stays NaN)
    const double T = tab[ki & 31];
    double p = __builtin_fma(r, 1.0 / 720.0, 1.0 / 120.0);
    p = __builtin_fma(r, p, 1.0 / 24.0);
    p = __builtin_fma(r, p, 1.0 / 6.0);
    p = __builtin_fma(r, p, 0.5);
    p = __builtin_fma(r, p, 1.0);
    p = r * p;
    return __builtin_amdgcn_ldexp(__builtin_fma(T, p, T), ki >> 5);
}

__device__ __forceinline__ double tab_log(double x, const double *__restrict__ tab) {   // x: a positive normal number
    constexpr double LN2 = 0x1.62e42fefa39efp-1;
    const double m = __builtin_amdgcn_frexp_mant(x);              // [0.5, 1)
    const int e = __builtin_amdgcn_frexp_exp(x);
    const int i = (__double2hiint(m) >> 15) & 31;                 // the five leading fraction bits
    const double r = __builtin_fma(m, tab[32 + i], -1.0);
    double p = __builtin_fma(r, -1.0 / 8.0, 1.0 / 7.0);
    p = __builtin_fma(r, p, -1.0 / 6.0);
    p = __builtin_fma(r, p, 1.0 / 5.0);
    p = __builtin_fma(r, p, -1.0 / 4.0);
    p = __builtin_fma(r, p, 1.0 / 3.0);
    p = __builtin_fma(r, p, -0.5);
    p = __builtin_fma(r, p, 1.0);
    return __builtin_fma((double)e, LN2, __builtin_fma(r, p, tab[64 + i]));
}

template <int NH, bool RCP, bool TAB>
__device__ __forceinline__ bool axis_gp(double v, const double *ax, double d, double noise_var, double ninv, double c1, double c2,
                                        double Q, const double *__restrict__ tab, double (&e)[1 << NH], double &sum) {
    constexpr int R = 1 << NH, JL = R / 2 - 1, JH = R / 2;
    constexpr int GL = JL ^ (JL >> 1), GH = JH ^ (JH >> 1);       // labels of the two middle levels
    // Round 5: TWO exp per axis.  With rho = e[JH] / e[JL] = exp((2 d (x - p[JL]) - d^2) / N0), the ratio between the two middle levels,
    // the upper middle level is e[JL] rho, the first ratio upwards rho Q and the first ratio downwards Q / rho: the second middle
    // exponential and both outward ratios of round 4's four-exp form follow from ONE exponential and one division (37 + 37 + 37
    // instructions -> 12 + 4).  rho itself must be a well-scaled normal number (it overflows only beyond Es/N0 ~ 33 dB, where the
    // middle exponential has long failed its own test); it joins the guard.
    const double dl = v - ax[GL];
    const double xl = RCP ? (dl * dl) * ninv : (-(dl * dl)) / noise_var;
    const double el = TAB ? tab_exp(xl, tab) : exp(xl);
    const double rho = TAB ? tab_exp(c1 * dl - c2, tab) : exp(c1 * dl - c2);
    const double eh = el * rho;
    e[GL] = el;
    e[GH] = eh;
    double cur = eh, r = rho * Q;
#pragma unroll
    for (int j = JH + 1; j < R; j++) { cur *= r; e[j ^ (j >> 1)] = cur; r *= Q; }
    cur = el;
    r = Q / rho;
#pragma unroll
    for (int j = JL - 1; j >= 0; j--) { cur *= r; e[j ^ (j >> 1)] = cur; r *= Q; }
    sum = 0.0;
#pragma unroll
    for (int a = 0; a < R; a++) sum += e[a];                      // label order, like the plain path
    return !(fmin(el, eh) >= 1e-290) || !(rho > 1e-290 && rho < 1e290) || !(sum >= 1e-30);
}

// ---- generic constellations (PSK, custom tables), M <= MAX_M: the fast form (round 6) ------------------------------------------------
// demod_soft_kernel above is the reference's formula instruction for instruction -- hypot, exp, a division per exponent, per LLR a
// division and a logarithm, and NB doubles per lane stored at an 8 NB-byte stride: ~870 vector instructions per 8-PSK symbol, a
// quarter of the HBM rate by arithmetic alone, and the store pattern that held the 64-QAM kernel at 36 % until round 5.  Here:
//   * |y - c|^2 = dx^2 + dy^2 (no hypot), the exponent by the reciprocal (RCP), exp / log from the 32-entry tables of the separable
//     kernel, the quotient by div_nr -- ~270 instructions per 8-PSK symbol; sums still in increasing constellation index (:128-136);
//   * the NB LLRs of a wave's 64 symbols leave as ONE contiguous run through a wave-private LDS tile (16 bytes per lane, consecutive
//     lanes consecutive addresses), any NB.  The 16-byte stores need a 16-byte aligned output array: a caller's odd pointer (a
//     sub-buffer at an odd element offset; round-5 advisor finding) is served by the literal kernel, whose stores are per element --
//     a run-time choice INSIDE the fast kernels cost the 64-QAM one 5 % (a store that exists on one path only is waited for on both);
//   * the same guard as the separable kernel: a symbol whose total is below 1e-290, or any of whose quotients leaves (e^-600, e^600)
//     or is not finite, is decided again by the literal formula, so the reference's -inf / NaN pattern and rounding near the
//     underflow range are kept (modulation.py:134-137).
// cpx_demod_set_path("libm") keeps the literal kernel for every symbol (the row this one is tested against).
template <int NB, bool RCP>
__global__ __launch_bounds__(DEMOD_BLOCK) void demod_soft_gen_kernel(const double2 *__restrict__ y, int64_t Ns,
                                                                     const double2 *__restrict__ cst, double noise_var, double scale,
                                                                     double *__restrict__ llr) {
    constexpr int M = 1 << NB;
    __shared__ double2 c_s[M];
    __shared__ double2 cn_s[RCP ? M : 1];                             // RCP: the points in units of sqrt(noise_var)
    __shared__ double tab_s[96];
    // RCP: symbol and points are scaled by 1 / sqrt(noise_var) once, so that the exponent -(dx^2 + dy^2) / noise_var is a
    // multiplication and a fused multiply-add per point instead of two multiplications, an addition and the scaling (the literal
    // redo below works on the unscaled values)
    const double rs = RCP ? sqrt(1.0 / noise_var) : 1.0;
    for (int m = threadIdx.x; m < M; m += DEMOD_BLOCK) {
        const double2 c = cst[m];
        c_s[m] = c;
        if (RCP) cn_s[m] = make_double2(c.x * rs, c.y * rs);
    }
    for (int m = threadIdx.x; m < 96; m += DEMOD_BLOCK) tab_s[m] = DEMOD_TAB[m];
    __syncthreads();
    double pend[NB];
    auto symbol = [&](const double2 cur) __attribute__((always_inline)) {
        double num[NB], den[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) { num[b] = 0.0; den[b] = 0.0; }
        const double ux = cur.x * rs, uy = cur.y * rs;
        auto point = [&](const int m) __attribute__((always_inline)) {
            double x;
            if (RCP) {
                const double2 c = cn_s[m];
                const double dx = ux - c.x, dy = uy - c.y;
                x = __builtin_fma(-dx, dx, -(dy * dy));
            } else {
                const double2 c = c_s[m];
                const double dx = cur.x - c.x, dy = cur.y - c.y;
                x = (-(dx * dx + dy * dy)) / noise_var;
            }
            const double e = tab_exp(x, tab_s);
#pragma unroll
            for (int b = 0; b < NB; b++) {
                if ((m >> b) & 1) num[b] += e; else den[b] += e;
            }
        };
        if constexpr (M <= 16) {                                      // unrolled: the bit tests are compile-time, a sum is one addition
#pragma unroll
            for (int m = 0; m < M; m++) point(m);
        } else {
#pragma unroll 8
            for (int m = 0; m < M; m++) point(m);
        }
        bool redo = !(num[0] + den[0] >= 1e-290);                     // every point's probability down among the denormals
        double out[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const double qa = div_nr(num[b], den[b]);
            redo |= !(qa > 2.7e-261 && qa < 3.7e260);
            out[b] = tab_log(qa, tab_s);
        }
        if (redo) {                                                   // the reference's way, as demod_soft_kernel
#pragma unroll
            for (int b = 0; b < NB; b++) { num[b] = 0.0; den[b] = 0.0; }
            for (int m = 0; m < M; m++) {
                const double2 c = c_s[m];
                const double a = hypot(cur.x - c.x, cur.y - c.y);
                const double e = exp((-(a * a)) / noise_var);
#pragma unroll
                for (int b = 0; b < NB; b++) {
                    if ((m >> b) & 1) num[b] += e; else den[b] += e;
                }
            }
#pragma unroll
            for (int b = 0; b < NB; b++) out[b] = fast_log(num[b] / den[b]);
        }
#pragma unroll
        for (int b = 0; b < NB; b++) pend[b] = out[b] * scale;        // (:137)
    };
    constexpr int WAVES = DEMOD_BLOCK / 64;
    __shared__ double xpose[WAVES * 64 * NB];
    const int lane = threadIdx.x & 63;
    double *tile = xpose + (threadIdx.x >> 6) * 64 * NB;
    auto flush = [&](const int64_t base, const bool full) __attribute__((always_inline)) {   // base: the wave's first symbol of that trip
#pragma unroll
        for (int b = 0; b < NB; b++) tile[lane * NB + NB - 1 - b] = pend[b];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int64_t total = Ns * NB, g0 = base * NB;
#pragma unroll
        for (int k = 0; k < (NB + 1) / 2; k++) {
            const int e = k * 128 + lane * 2;
            if (e < 64 * NB) {                                        // (odd NB: the last store instruction is half a wave)
                const double2 v = *reinterpret_cast<const double2 *>(tile + e);
                if (full || g0 + e + 1 < total) *reinterpret_cast<double2 *>(llr + g0 + e) = v;
                else if (g0 + e < total) llr[g0 + e] = v.x;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // the software pipeline of demod_soft_sep_kernel: next symbol requested and previous LLRs stored at the top of a trip
    const int64_t stride = (int64_t)gridDim.x * DEMOD_BLOCK;
    int64_t base = (int64_t)blockIdx.x * DEMOD_BLOCK + (threadIdx.x & ~63);     // wave-uniform
    if (base >= Ns) return;
    auto at = [&](int64_t wb) { return y[wb + lane < Ns ? wb + lane : Ns - 1]; };
    double2 cur = at(base);
    double2 nxt = at(base + stride < Ns ? base + stride : base);
    symbol(cur);
    int64_t prev = base;
    for (base += stride; base < Ns; base += stride) {
        cur = nxt;
        flush(prev, true);
        nxt = at(base + stride < Ns ? base + stride : base);
        symbol(cur);
        prev = base;
    }
    flush(prev, prev + 64 <= Ns);
}

// One symbol of an axis-separable constellation -> its NB = 2 NH LLRs (label bit b in out[b], unscaled): the arithmetic of
// demod_soft_sep_kernel, shared with link_front_kernel (the transmit chain + channel fused in front of it, round 6).
struct SepCtx {
    const double *ax, *tab;                                        // LDS: [xs | ys] levels; exp / log tables (TAB)
    double noise_var, ninv, step_x, step_y, c1x, c2x, Qx, c1y, c2y, Qy;
};

template <bool GP>
__device__ __forceinline__ SepCtx sep_ctx(const double *ax, const double *tab, double noise_var, double step_x, double step_y) {
    SepCtx c;
    c.ax = ax; c.tab = tab; c.noise_var = noise_var; c.ninv = -1.0 / noise_var; c.step_x = step_x; c.step_y = step_y;
    // GP: per-call constants of the two axes (2 d / N0, d^2 / N0, Q)
    c.c1x = GP ? 2.0 * step_x / noise_var : 0.0; c.c2x = GP ? step_x * step_x / noise_var : 0.0; c.Qx = GP ? exp(-2.0 * c.c2x) : 0.0;
    c.c1y = GP ? 2.0 * step_y / noise_var : 0.0; c.c2y = GP ? step_y * step_y / noise_var : 0.0; c.Qy = GP ? exp(-2.0 * c.c2y) : 0.0;
    return c;
}

template <int NH, bool RCP, bool GP, bool TAB>
__device__ __forceinline__ void sep_symbol(const SepCtx &c, const double2 cur, double (&out)[2 * NH]) {
    constexpr int R = 1 << NH, NB = 2 * NH;
    double ex[R], ey[R], sx = 0.0, sy = 0.0;
    bool redo = false;
    if constexpr (GP && NH == 1) {
        // Two levels per axis (QPSK / 4-QAM): LLR(bit of the real axis) = log(e[1] sy / (e[0] sy)) = (dx0^2 - dx1^2) / N0 -- no
        // exp, no log, no division; the kernel is then bound by its 32 bytes per symbol.  Valid while every one of the four
        // point probabilities exp(-(dx^2 + dy^2) / N0) the reference adds up is a normal number with room to spare: all four
        // exponents below 650 (e^-650 = 1e-282).  Beyond that -- far outliers, Es/N0 above ~22 dB -- the reference's sums
        // underflow in its own pattern (-inf, NaN) and the symbol is decided point by point below, as before.
        const double dx0 = cur.x - c.ax[0], dx1 = cur.x - c.ax[1], dy0 = cur.y - c.ax[R], dy1 = cur.y - c.ax[R + 1];
        const double qx0 = dx0 * dx0, qx1 = dx1 * dx1, qy0 = dy0 * dy0, qy1 = dy1 * dy1;
        const double worst = RCP ? (fmax(qx0, qx1) + fmax(qy0, qy1)) * -c.ninv : (fmax(qx0, qx1) + fmax(qy0, qy1)) / c.noise_var;
        out[1] = RCP ? (qx1 - qx0) * c.ninv : (qx0 - qx1) / c.noise_var;           // label bit 1 = real-axis index
        out[0] = RCP ? (qy1 - qy0) * c.ninv : (qy0 - qy1) / c.noise_var;           // label bit 0 = imag-axis index
        if (!(worst < 650.0)) {
            double num[NB] = {0.0, 0.0}, den[NB] = {0.0, 0.0};
            for (int m = 0; m < R * R; m++) {
                const double h = hypot(cur.x - c.ax[m >> NH], cur.y - c.ax[R + (m & (R - 1))]);
                const double e = exp((-(h * h)) / c.noise_var);
#pragma unroll
                for (int b = 0; b < NB; b++) {
                    if ((m >> b) & 1) num[b] += e; else den[b] += e;
                }
            }
#pragma unroll
            for (int b = 0; b < NB; b++) out[b] = fast_log(num[b] / den[b]);
        }
        return;
    }
    if (GP) {
        redo |= axis_gp<NH, RCP, TAB>(cur.x, c.ax, c.step_x, c.noise_var, c.ninv, c.c1x, c.c2x, c.Qx, c.tab, ex, sx);
        redo |= axis_gp<NH, RCP, TAB>(cur.y, c.ax + R, c.step_y, c.noise_var, c.ninv, c.c1y, c.c2y, c.Qy, c.tab, ey, sy);
    } else {
#pragma unroll
        for (int a = 0; a < R; a++) {
            const double dx = cur.x - c.ax[a], dy = cur.y - c.ax[R + a];
            ex[a] = exp(RCP ? (dx * dx) * c.ninv : (-(dx * dx)) / c.noise_var);
            ey[a] = exp(RCP ? (dy * dy) * c.ninv : (-(dy * dy)) / c.noise_var);
            sx += ex[a];
            sy += ey[a];
        }
    }
#pragma unroll
    for (int b = 0; b < NH; b++) {
        double nx = 0.0, qx = 0.0, ny = 0.0, qy = 0.0;
#pragma unroll
        for (int a = 0; a < R; a++) {
            if ((a >> b) & 1) { nx += ex[a]; ny += ey[a]; } else { qx += ex[a]; qy += ey[a]; }
        }
        // label bit NH+b = bit b of the real-axis index a, label bit b = bit b of the imag-axis index.  Round 4: the quotient by
        // reciprocal + Newton and the logarithm without its special-value selects (cpx_math.h: -20 instructions per LLR) -- the
        // symbols those would be needed for (a quotient outside (e^-600, e^600), zero, inf or NaN: exactly the |LLR| >= 600 /
        // non-finite rule below, tested on the quotient) are decided point by point anyway
        const double qa = div_nr(nx * sy, qx * sy), qb = div_nr(ny * sx, qy * sx);
        redo |= !(qa > 2.7e-261 && qa < 3.7e260) || !(qb > 2.7e-261 && qb < 3.7e260);
        out[NH + b] = TAB ? tab_log(qa, c.tab) : fast_log<false, true>(qa);
        out[b] = TAB ? tab_log(qb, c.tab) : fast_log<false, true>(qb);
    }
    // The factorised sums are only as good as the reference's point-by-point ones while nothing is near the underflow
    // threshold: an LLR beyond +-600 (or non-finite) means some sum of e^{-d^2/N0} terms is down among the denormals, where
    // sum-of-products and product-of-sums round differently (measured at 29 dB: 0.9 apart at |LLR| = 740, a handful of
    // inf / finite flips).  Such a symbol is redone the reference's way -- every point, hypot, division, increasing label --
    // so that its rounding and its +-inf / NaN pattern are the reference's (modulation.py:125-137).
#pragma unroll
    for (int b = 0; b < NB; b++) redo |= !(fabs(out[b]) < 600.0);
    if (redo) {
        double num[NB], den[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) { num[b] = 0.0; den[b] = 0.0; }
        for (int m = 0; m < R * R; m++) {
            const double h = hypot(cur.x - c.ax[m >> NH], cur.y - c.ax[R + (m & (R - 1))]);
            const double e = exp((-(h * h)) / c.noise_var);
#pragma unroll
            for (int b = 0; b < NB; b++) {
                if ((m >> b) & 1) num[b] += e; else den[b] += e;
            }
        }
#pragma unroll
        for (int b = 0; b < NB; b++) out[b] = fast_log(num[b] / den[b]);
    }
}

template <int NH, bool RCP, bool GP, bool TAB = false>
__global__ __launch_bounds__(DEMOD_BLOCK) void demod_soft_sep_kernel(const double2 *__restrict__ y, int64_t Ns,
                                                                     const double *__restrict__ axes, double noise_var,
                                                                     double scale, double step_x, double step_y,
                                                                     double *__restrict__ llr) {
    constexpr int R = 1 << NH, NB = 2 * NH;
    __shared__ double ax_s[2 * R];
    __shared__ double tab_s[TAB ? 96 : 1];
    for (int m = threadIdx.x; m < 2 * R; m += DEMOD_BLOCK) ax_s[m] = axes[m];
    if (TAB)
        for (int m = threadIdx.x; m < 96; m += DEMOD_BLOCK) tab_s[m] = DEMOD_TAB[m];
    __syncthreads();
    const SepCtx ctx = sep_ctx<GP>(ax_s, tab_s, noise_var, step_x, step_y);
    // One symbol: cur -> NB LLRs in `pend` (two levels per axis: stored at once).
    constexpr bool DEFER = !(GP && NH == 1);
    double pend[DEFER ? NB : 1];
    auto symbol = [&](const double2 cur, const int64_t i) __attribute__((always_inline)) {
        double out[NB];
        sep_symbol<NH, RCP, GP, TAB>(ctx, cur, out);
        if constexpr (!DEFER) {
#pragma unroll
            for (int b = 0; b < NB; b++)
                if (i < Ns) llr[i * NB + NB - 1 - b] = out[b] * scale;                // (:137)
        } else {
#pragma unroll
            for (int b = 0; b < NB; b++) pend[b] = out[b] * scale;                    // (:137)
        }
    };
    // The NB LLRs of a wave's 64 consecutive symbols are one contiguous run of 64 NB doubles.  Stored lane by lane (16 bytes per lane at
    // a 48-byte stride for 64-QAM) each store instruction touched 48 cache lines, and the kernel ran at the speed of that pattern alone:
    // 0.212 ms with the arithmetic removed, 0.123 ms with the stores removed (profiles/r05_demod_ablation.txt).  The wave transposes
    // through a private LDS tile instead and every store instruction writes 1 KB of consecutive bytes.
    constexpr int WAVES = DEMOD_BLOCK / 64;
    __shared__ double xpose[DEFER ? WAVES * 64 * NB : 1];
    const int lane = threadIdx.x & 63;
    double *tile = xpose + (DEFER ? (threadIdx.x >> 6) * 64 * NB : 0);
    auto flush = [&](const int64_t base, const bool full) __attribute__((always_inline)) {   // base: the wave's first symbol of that trip
        if (DEFER) {
#pragma unroll
            for (int b = 0; b < NB; b++) tile[lane * NB + NB - 1 - b] = pend[DEFER ? b : 0];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int64_t total = Ns * NB, g0 = base * NB;
#pragma unroll
            for (int k = 0; k < NB / 2; k++) {
                const int e = k * 128 + lane * 2;
                const double2 v = *reinterpret_cast<const double2 *>(tile + e);
                if (full || g0 + e + 1 < total) *reinterpret_cast<double2 *>(llr + g0 + e) = v;   // (16-byte aligned: the host sends a
                else if (g0 + e < total) llr[g0 + e] = v.x;                                         //  caller's odd pointer to demod_soft_kernel)
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    };
    // Software pipeline over a wave's symbols (round 5).  The load sat at the top of the loop body with its wait right behind it and
    // the stores at the bottom, in front of the loop header's vmcnt(0).  Now the next symbol is requested and the previous trip's LLRs
    // are stored at the TOP of a trip -- both complete while the current symbol is computed.  The first trip is peeled off (a store
    // that is there on one path into the loop header and not on the other makes the compiler wait for the worst case on both), and
    // the trip count is the WAVE's: lanes past the end of the input work on a copy of the last symbol and store nothing.
    const int64_t stride = (int64_t)gridDim.x * DEMOD_BLOCK;
    int64_t base = (int64_t)blockIdx.x * DEMOD_BLOCK + (threadIdx.x & ~63);     // wave-uniform
    if (base >= Ns) return;
    auto at = [&](int64_t wb) { return y[wb + lane < Ns ? wb + lane : Ns - 1]; };
    double2 cur = at(base);
    double2 nxt = at(base + stride < Ns ? base + stride : base);
    symbol(cur, base + lane);
    int64_t prev = base;
    for (base += stride; base < Ns; base += stride) {
        cur = nxt;
        flush(prev, true);                                       // only a wave's LAST trip can be ragged: the stores of this one are
        nxt = at(base + stride < Ns ? base + stride : base);     // unconditional, their number known where the next wait is placed
        symbol(cur, base + lane);
        prev = base;
    }
    flush(prev, prev + 64 <= Ns);
}

// ---- the transmit chain and the channel fused in front of the separable soft demodulator (round 6) ------------------------
// A Monte-Carlo link point used to be seven element-wise kernels -- random message bits, conv_encode, puncturing gather, modulate,
// AWGN, soft demodulation, depuncturing gather (linksim.hip, devicelink.py) -- each writing an array the next one read back:
// ~65 KB of HBM traffic per 1200-bit MCS-5 frame for 20.4 KB of results, 2.2 of the 5.2 ms of a BASELINE config-5 sweep.  Every stage
// is a pure function of (frame, position) and of counter-based random numbers (cpx_rng.h), so ONE kernel can produce a symbol's LLRs
// from nothing but its index: the lane of symbol (f, s)
//   * regenerates the message bits its coded bits depend on (16 bits per Philox counter value: at most four values cover the window
//     of a symbol -- checked when the plan is made) and stores the message bits it owns, [s nbits / nsym, (s + 1) nbits / nsym);
//   * walks the feed-forward encoder table for the NB coded positions the puncturing keeps for it (convcode.py:531-550, :752-774),
//     MSB-first label -> the constellation point (modulation.py:93-96), adds the noise of element f nsym + s (channels.py:37-55);
//   * runs sep_symbol -- the arithmetic of demod_soft_sep_kernel, same template arguments the staged path would use (modulation.py:
//     125-137);
//   * the wave's 64 NB LLRs go through the LDS tile and leave in transmitted-bit order, each to its decoder-input position with
//     the punctured positions behind it set to 0.0 (convcode.py:777-804): consecutive lanes, near-consecutive addresses.
// The values are those of the staged kernels bit for bit (same generators, same streams, same arithmetic;
// tests/test_devicelink_gpu.py::test_fused_front_end_equals_staged_chain); the only HBM traffic left is the results.
struct LinkFrontParams {
    int32_t T;                                                     // transmissions (T nsym < 2^31)
    int nbits, mem;                                                // message bits per transmission; encoder memory
    int ntx, nsym, nde;                                            // transmitted bits, symbols, decoder inputs per transmission
    const int32_t *jo, *pg;                                        // [ntx], per transmitted bit t: (trellis step << 4) | output bit counted from the
                                                                   // MSB of the code word;  decoder-input position | (punctured positions behind it << 26)
    int lead;                                                      // punctured decoder-input positions in front of transmitted bit 0
    const int32_t *out_tab;                                        // [2^mem][2] encoder outputs
    const double *axes;
    double step_x, step_y, noise_var, scale_re, scale_im, llr_scale;
    uint64_t seed, stream_bits, stream_noise;
    uint8_t *msg;
    double *llr;
    double2 *rx;                                                   // optional: the noisy symbols (tests)
};

// (four waves per SIMD for up to 64-QAM: 128 VGPRs and six spilled values instead of 147 -- front end of the config-5 sweep 0.985 -> 0.94 ms,
//  same box, alternating; five / six waves spill 38 / 57 values and lose: 1.36 / 1.69 ms)
template <int NH, bool RCP, bool GP, bool TAB>
__global__ __launch_bounds__(DEMOD_BLOCK, (NH <= 3 ? 4 : 3)) void link_front_kernel(const LinkFrontParams p) {
    constexpr int R = 1 << NH, NB = 2 * NH, WAVES = DEMOD_BLOCK / 64;
    __shared__ double ax_s[2 * R];
    __shared__ double tab_s[TAB ? 96 : 1];
    __shared__ int32_t out_s[128];
    __shared__ double xpose[WAVES * 64 * NB];
    for (int m = threadIdx.x; m < 2 * R; m += DEMOD_BLOCK) ax_s[m] = p.axes[m];
    if (TAB)
        for (int m = threadIdx.x; m < 96; m += DEMOD_BLOCK) tab_s[m] = DEMOD_TAB[m];
    for (int m = threadIdx.x; m < (2 << p.mem); m += DEMOD_BLOCK) out_s[m] = p.out_tab[m];
    __syncthreads();
    const SepCtx ctx = sep_ctx<GP>(ax_s, tab_s, p.noise_var, p.step_x, p.step_y);
    const int lane = threadIdx.x & 63;
    double *tile = xpose + (threadIdx.x >> 6) * 64 * NB;
    const uint32_t Ns = (uint32_t)p.T * (uint32_t)p.nsym, stride = gridDim.x * DEMOD_BLOCK;
    const int nbits = p.nbits, mem = p.mem, nsym = p.nsym;
    const uint32_t fmask = (2u << mem) - 1u, smask = (1u << mem) - 1u;
    for (uint32_t base = blockIdx.x * DEMOD_BLOCK + (threadIdx.x & ~63u); base < Ns; base += stride) {
        // the wave's first symbol: transmission and position by one scalar division, the lanes count on from there
        const uint32_t wbase = __builtin_amdgcn_readfirstlane(base);
        const uint32_t f0 = wbase / (uint32_t)nsym;
        const int s0 = (int)(wbase - f0 * (uint32_t)nsym);
        const bool live = wbase + lane < Ns;
        uint32_t f = f0;
        int s = s0 + (live ? lane : 0);
        while (s >= nsym) { s -= nsym; f++; }
        // ---- message window: bit (j - lo) of `win` = message bit j of this transmission, 0 outside [0, nbits) ----
        int jo[NB];
#pragma unroll
        for (int q = 0; q < NB; q++) jo[q] = p.jo[s * NB + q];
        const int o0 = (int)((uint32_t)s * (uint32_t)nbits / (uint32_t)nsym), o1 = (int)(((uint32_t)s + 1u) * (uint32_t)nbits / (uint32_t)nsym);
        int lo = (jo[0] >> 4) - mem, hi = jo[NB - 1] >> 4;
        lo = lo < o0 ? lo : o0;
        hi = hi > o1 - 1 ? hi : o1 - 1;
        hi = hi > nbits - 1 ? nbits - 1 : hi;
        const int lo0 = lo < 0 ? 0 : lo;                           // first real message bit of the window
        const int64_t fbit = (int64_t)f * nbits;                   // index of the transmission's first message bit in the point's stream
        const int64_t w0 = (fbit + lo0) >> 4, w1 = (fbit + hi) >> 4;
        uint64_t win = 0;
        for (int64_t w = w0; w <= w1; w++) win |= (uint64_t)message_bits16((uint64_t)w, p.stream_bits, p.seed) << (16 * (int)(w - w0));
        win >>= (int)(fbit + lo0 - (w0 << 4));                      // bit 0 = message bit lo0
        const int valid = hi - lo0 + 1;                            // (1 .. 49: checked when the plan was made)
        win &= (~0ull) >> (64 - valid);
        win <<= lo0 - lo;                                          // zeros for the positions in front of the transmission
        if (live)
            for (int j = o0; j < o1; j++) p.msg[fbit + j] = (uint8_t)((win >> (j - lo)) & 1u);
        // ---- encoder at the kept positions (state = the previous mem bits, most recent = MSB), label, constellation point, noise ----
        int label = 0;
#pragma unroll
        for (int q = 0; q < NB; q++) {
            const int j = jo[q] >> 4;
            const uint32_t fld = (uint32_t)(win >> (j - mem - lo)) & fmask;       // bits j - mem .. j
            const int code = out_s[(fld & smask) * 2 + (fld >> mem)];
            label = (label << 1) | ((code >> (jo[q] & 15)) & 1);                   // dec2bitarray(o, n), MSB first (:535)
        }
        const double2 x = make_double2(ax_s[label >> NH], ax_s[R + (label & (R - 1))]);
        const double2 y = awgn_add(x, (uint64_t)f * (uint64_t)nsym + (uint64_t)s, p.scale_re, p.scale_im, p.seed, p.stream_noise);
        if (p.rx && live) p.rx[wbase + lane] = y;
        // ---- LLRs ----
        double out[NB];
        sep_symbol<NH, RCP, GP, TAB>(ctx, y, out);
#pragma unroll
        for (int b = 0; b < NB; b++) tile[lane * NB + NB - 1 - b] = out[b] * p.llr_scale;   // (:137)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- transmitted-bit order -> decoder-input positions, zeros behind (and, for bit 0, in front of) the kept ones ----
        const uint32_t left = (Ns - wbase) * NB;                    // transmitted bits from the wave's first one to the end (>= 1)
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const uint32_t e = k * 64 + lane;
            if (e < left) {
                uint32_t ff = f0;
                int t = s0 * NB + (int)e;
                while (t >= p.ntx) { t -= p.ntx; ff++; }
                const int pg = p.pg[t];
                const int P = pg & 0x3FFFFFF, gap = (int)((uint32_t)pg >> 26);
                double *o = p.llr + (int64_t)ff * p.nde + P;
                o[0] = tile[e];
                for (int z = 1; z <= gap; z++) o[z] = 0.0;
                if (t == 0)
                    for (int z = 1; z <= p.lead; z++) o[-z] = 0.0;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// ---- "fp32-fast" soft decisions (cpx_set_precision) ---------------------------------------------------------------------
// Not the parity mode: float32 arithmetic with the hardware exp2 / log2, in log-sum-exp form -- every exponent is taken relative to
// the largest of its sum, so nothing underflows where float64 would not either and the result is FINITE wherever the exponents are
// (the reference's -inf / NaN pattern of modulation.py:134-137 at extreme Es/N0 is deliberately not reproduced; it yields the LLR the
// formula defines).  Contract (tests/test_fp32_fast_gpu.py): |LLR_fast - LLR_fp64| <= 2e-5 + 4e-6 |LLR_fp64| (measured: a quarter of
// that at worst, PSK-2..16 / QAM-4..256, Es/N0 0..28 dB), identical hard decisions sign(LLR) wherever |LLR_fp64| > 1e-3.  Inputs and outputs stay float64 arrays (the API does not
// change); the kernel is then bound by its 16 + 8 nb bytes per symbol, so the LLRs of a wave go through LDS and leave as full 512-byte
// rows (a lane's own nb doubles are 8 nb bytes apart from its neighbour's).
constexpr float LN2_F = 0.6931471805599453f;
// a sum below this is re-formed relative to its own largest term (v_exp_f32 / v_log_f32 lose their accuracy among the denormals)
constexpr float F32_TINY = 1e-30f;

// log2( sum_{a in sel} 2^t[a] ) for the members of `t` whose label has bit b set (WANT = 1) or clear (WANT = 0)
template <int R, int WANT>
__device__ __forceinline__ float lse2_subset(const float (&t)[R], int b) {
    float m = -INFINITY;
#pragma unroll
    for (int a = 0; a < R; a++) if (((a >> b) & 1) == WANT) m = fmaxf(m, t[a]);
    float s = 0.0f;
#pragma unroll
    for (int a = 0; a < R; a++) if (((a >> b) & 1) == WANT) s += __builtin_amdgcn_exp2f(t[a] - m);
    return m + __builtin_amdgcn_logf(s);                          // v_log_f32 is log2
}

template <int NB>
__device__ __forceinline__ void store_rows_f32(const float (&out)[NB], float scale, int64_t i0, int64_t Ns, double *__restrict__ llr,
                                               float *stage) {
    // stage: [64 * NB] floats of this wave; lane l holds symbol i0 + l; out[b] goes to llr[(i0 + l) * NB + NB - 1 - b]
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int b = 0; b < NB; b++) stage[lane * NB + NB - 1 - b] = out[b] * scale;
    __builtin_amdgcn_wave_barrier();
    const int64_t base = i0 * NB, end = Ns * NB;
#pragma unroll
    for (int k = 0; k < NB; k++) {
        const int64_t o = base + k * 64 + lane;
        if (o < end) llr[o] = (double)stage[k * 64 + lane];
    }
    __builtin_amdgcn_wave_barrier();
}

template <int NH>
__global__ __launch_bounds__(DEMOD_BLOCK) void demod_soft_sep_f32_kernel(const double2 *__restrict__ y, int64_t Ns,
                                                                         const double *__restrict__ axes, double noise_var,
                                                                         double scale, double *__restrict__ llr) {
    constexpr int R = 1 << NH, NB = 2 * NH;
    __shared__ float ax_s[2 * R];
    __shared__ float stage_s[DEMOD_BLOCK * NB];
    for (int m = threadIdx.x; m < 2 * R; m += DEMOD_BLOCK) ax_s[m] = (float)axes[m];
    __syncthreads();
    const float c = (float)(-1.4426950408889634 / noise_var);     // exponents in base 2
    const float sc = (float)scale * LN2_F;
    float *stage = stage_s + (threadIdx.x >> 6) * 64 * NB;
    // whole waves walk the array together (the row stores need all 64 lanes of a wave at the same trip count)
    const int64_t n_rows = (Ns + 63) >> 6, wave0 = (int64_t)blockIdx.x * (DEMOD_BLOCK / 64) + (threadIdx.x >> 6);
    for (int64_t row = wave0; row < n_rows; row += (int64_t)gridDim.x * (DEMOD_BLOCK / 64)) {
        const int64_t i0 = row << 6, i = i0 + (threadIdx.x & 63);
        const double2 cur = y[i < Ns ? i : Ns - 1];
        const float x = (float)cur.x, yy = (float)cur.y;
        float tx[R], ty[R];
#pragma unroll
        for (int a = 0; a < R; a++) {
            const float dx = x - ax_s[a], dy = yy - ax_s[R + a];
            tx[a] = dx * dx * c;
            ty[a] = dy * dy * c;
        }
        // R exponentials per axis relative to the axis' nearest level; a bit subset whose members all flush to zero against that
        // maximum (|LLR| beyond ~69) is summed again relative to its own maximum
        float mx = tx[0], my = ty[0];
#pragma unroll
        for (int a = 1; a < R; a++) { mx = fmaxf(mx, tx[a]); my = fmaxf(my, ty[a]); }
        float ex[R], ey[R];
#pragma unroll
        for (int a = 0; a < R; a++) { ex[a] = __builtin_amdgcn_exp2f(tx[a] - mx); ey[a] = __builtin_amdgcn_exp2f(ty[a] - my); }
        float out[NB];
#pragma unroll
        for (int b = 0; b < NH; b++) {
            // label bit NH+b = bit b of the real-axis index, label bit b = bit b of the imag-axis index; the other axis cancels
            float nx = 0.0f, qx = 0.0f, ny = 0.0f, qy = 0.0f;
#pragma unroll
            for (int a = 0; a < R; a++) {
                if ((a >> b) & 1) { nx += ex[a]; ny += ey[a]; } else { qx += ex[a]; qy += ey[a]; }
            }
            out[NH + b] = __builtin_amdgcn_logf(nx) - __builtin_amdgcn_logf(qx);
            out[b] = __builtin_amdgcn_logf(ny) - __builtin_amdgcn_logf(qy);
            if (!(nx >= F32_TINY) || !(qx >= F32_TINY)) out[NH + b] = lse2_subset<R, 1>(tx, b) - lse2_subset<R, 0>(tx, b);
            if (!(ny >= F32_TINY) || !(qy >= F32_TINY)) out[b] = lse2_subset<R, 1>(ty, b) - lse2_subset<R, 0>(ty, b);
        }
        store_rows_f32<NB>(out, sc, i0, Ns, llr, stage);
    }
}

// generic constellation (PSK, custom tables), <= MAX_M points in LDS: two scans per bit subset would cost 2 M nb exponentials, so one
// scan finds the nearest point overall, a second adds 2^(t - t_max) into the per-bit sums; a subset whose every member is 2^-126
// below the overall maximum (its sum flushes to zero) is evaluated again relative to its own maximum
template <int NB>
__global__ __launch_bounds__(DEMOD_BLOCK) void demod_soft_f32_kernel(const double2 *__restrict__ y, int64_t Ns,
                                                                     const double2 *__restrict__ cst, int M, double noise_var,
                                                                     double scale, double *__restrict__ llr) {
    __shared__ float2 c_s[MAX_M];
    __shared__ float stage_s[DEMOD_BLOCK * NB];
    for (int m = threadIdx.x; m < M; m += DEMOD_BLOCK) c_s[m] = make_float2((float)cst[m].x, (float)cst[m].y);
    __syncthreads();
    const float c = (float)(-1.4426950408889634 / noise_var);
    const float sc = (float)scale * LN2_F;
    float *stage = stage_s + (threadIdx.x >> 6) * 64 * NB;
    const int64_t n_rows = (Ns + 63) >> 6, wave0 = (int64_t)blockIdx.x * (DEMOD_BLOCK / 64) + (threadIdx.x >> 6);
    for (int64_t row = wave0; row < n_rows; row += (int64_t)gridDim.x * (DEMOD_BLOCK / 64)) {
        const int64_t i0 = row << 6, i = i0 + (threadIdx.x & 63);
        const double2 cur = y[i < Ns ? i : Ns - 1];
        const float x = (float)cur.x, yy = (float)cur.y;
        float tmax = -INFINITY;
        for (int m = 0; m < M; m++) {
            const float dx = x - c_s[m].x, dy = yy - c_s[m].y;
            tmax = fmaxf(tmax, (dx * dx + dy * dy) * c);
        }
        float num[NB], den[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) { num[b] = 0.0f; den[b] = 0.0f; }
        for (int m = 0; m < M; m++) {
            const float dx = x - c_s[m].x, dy = yy - c_s[m].y;
            const float e = __builtin_amdgcn_exp2f((dx * dx + dy * dy) * c - tmax);
#pragma unroll
            for (int b = 0; b < NB; b++) {
                if ((m >> b) & 1) num[b] += e; else den[b] += e;
            }
        }
        float out[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) {
            out[b] = __builtin_amdgcn_logf(num[b]) - __builtin_amdgcn_logf(den[b]);
            if (!(num[b] >= F32_TINY) || !(den[b] >= F32_TINY)) {   // a subset far below the nearest point: relative to its own maximum
                float mx[2] = {-INFINITY, -INFINITY}, s[2] = {0.0f, 0.0f};
                for (int m = 0; m < M; m++) {
                    const float dx = x - c_s[m].x, dy = yy - c_s[m].y;
                    const int w = (m >> b) & 1;
                    mx[w] = fmaxf(mx[w], (dx * dx + dy * dy) * c);
                }
                for (int m = 0; m < M; m++) {
                    const float dx = x - c_s[m].x, dy = yy - c_s[m].y;
                    const int w = (m >> b) & 1;
                    s[w] += __builtin_amdgcn_exp2f((dx * dx + dy * dy) * c - mx[w]);
                }
                out[b] = (mx[1] + __builtin_amdgcn_logf(s[1])) - (mx[0] + __builtin_amdgcn_logf(s[0]));
            }
        }
        store_rows_f32<NB>(out, sc, i0, Ns, llr, stage);
    }
}

__global__ __launch_bounds__(DEMOD_BLOCK) void demod_hard_kernel(const double2 *__restrict__ y, int64_t Ns,
                                                                 const double2 *__restrict__ cst, int M, int nb,
                                                                 int8_t *__restrict__ bits) {
    __shared__ double2 c_s[MAX_M];
    for (int m = threadIdx.x; m < M; m += DEMOD_BLOCK) c_s[m] = cst[m];
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * DEMOD_BLOCK + threadIdx.x; i < Ns; i += (int64_t)gridDim.x * DEMOD_BLOCK) {
        const int best = hard_scan(c_s, M, y[i]);
        for (int b = 0; b < nb; b++) bits[i * nb + b] = (int8_t)((best >> (nb - 1 - b)) & 1);   // dec2bitarray (:123)
    }
}

template <int NH>
__global__ __launch_bounds__(DEMOD_BLOCK) void demod_hard_sep_kernel(const double2 *__restrict__ y, int64_t Ns,
                                                                     const double2 *__restrict__ cst,
                                                                     const double *__restrict__ axes,
                                                                     int8_t *__restrict__ bits) {
    constexpr int R = 1 << NH, NB = 2 * NH, M = R * R;
    __shared__ double2 c_s[M];
    __shared__ double ax_s[2 * R];
    for (int m = threadIdx.x; m < M; m += DEMOD_BLOCK) c_s[m] = cst[m];
    for (int m = threadIdx.x; m < 2 * R; m += DEMOD_BLOCK) ax_s[m] = axes[m];
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * DEMOD_BLOCK + threadIdx.x; i < Ns; i += (int64_t)gridDim.x * DEMOD_BLOCK) {
        const int best = hard_sep<NH>(c_s, ax_s, y[i]);
        // dec2bitarray (:123): NB bytes of 0 / 1, MSB first, as ONE store per lane where the alignment allows it (round 4: six
        // byte stores per 64-QAM symbol left the kernel at 43 % of HBM peak); byte k of the little-endian word = bit NB-1-k
        unsigned long long w = 0;
#pragma unroll
        for (int b = 0; b < NB; b++) w |= (unsigned long long)((best >> (NB - 1 - b)) & 1) << (8 * b);
        int8_t *o = bits + i * NB;
        if (((uintptr_t)bits & 7) != 0) {                         // a caller's odd output pointer: byte stores
#pragma unroll
            for (int b = 0; b < NB; b++) o[b] = (int8_t)(w >> (8 * b));
        } else if constexpr (NB == 2) {
            *reinterpret_cast<unsigned short *>(o) = (unsigned short)w;                      // 2 i: 2-byte aligned
        } else if constexpr (NB == 4) {
            *reinterpret_cast<unsigned *>(o) = (unsigned)w;                                  // 4 i
        } else if constexpr (NB == 8) {
            *reinterpret_cast<unsigned long long *>(o) = w;                                  // 8 i
        } else {                                                                             // 6 i: 2-byte aligned
            *reinterpret_cast<unsigned short *>(o) = (unsigned short)w;
            *reinterpret_cast<unsigned short *>(o + 2) = (unsigned short)(w >> 16);
            *reinterpret_cast<unsigned short *>(o + 4) = (unsigned short)(w >> 32);
        }
    }
}

// ---- constellations of more than MAX_M points (the reference takes any power of two, modulation.py:159-166) -------------------
// The same formulas with the table read from HBM (wave-uniform addresses: scalar loads) and up to 16 bits per symbol.
// Round 5: the exponent is ALWAYS the reference's own division (-(a^2)) / noise_var (round 4 multiplied by a reciprocal here without
// the "|LLR| >= 600 or non-finite -> redo point by point" rule of the LDS kernels, so near the underflow range the +-inf / NaN pattern
// and the last bits could differ from modulation.py:134-137), and the kernel is instantiated per bit count (only NB accumulator
// pairs exist; round 4 kept 16 pairs for every nb).  A completeness path: O(M) exponentials per symbol.
template <int NB>
__global__ __launch_bounds__(DEMOD_BLOCK) void demod_soft_any_kernel(const double2 *__restrict__ y, int64_t Ns,
                                                                     const double2 *__restrict__ cst, int M,
                                                                     double noise_var, double scale, double *__restrict__ llr) {
    for (int64_t i = (int64_t)blockIdx.x * DEMOD_BLOCK + threadIdx.x; i < Ns; i += (int64_t)gridDim.x * DEMOD_BLOCK) {
        const double2 cur = y[i];
        double num[NB], den[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) { num[b] = 0.0; den[b] = 0.0; }
        for (int m = 0; m < M; m++) {
            const double2 c = cst[m];
            const double a = hypot(cur.x - c.x, cur.y - c.y);       // abs(current_symbol - symbol)
            const double e = exp((-(a * a)) / noise_var);           // (:134,136)
#pragma unroll
            for (int b = 0; b < NB; b++) {
                if ((m >> b) & 1) num[b] += e; else den[b] += e;
            }
        }
#pragma unroll
        for (int b = 0; b < NB; b++) llr[i * NB + NB - 1 - b] = fast_log(num[b] / den[b]) * scale;   // (:137)
    }
}

__global__ __launch_bounds__(DEMOD_BLOCK) void demod_hard_any_kernel(const double2 *__restrict__ y, int64_t Ns,
                                                                     const double2 *__restrict__ cst, int M, int nb,
                                                                     int8_t *__restrict__ bits) {
    for (int64_t i = (int64_t)blockIdx.x * DEMOD_BLOCK + threadIdx.x; i < Ns; i += (int64_t)gridDim.x * DEMOD_BLOCK) {
        const int best = hard_scan(cst, M, y[i]);
        for (int b = 0; b < nb; b++) bits[i * nb + b] = (int8_t)((best >> (nb - 1 - b)) & 1);   // dec2bitarray (:123)
    }
}

// "plain": the R-exponentials-per-axis form of the separable kernel also where the progression applies (A/B runs, tests);
// initial value from the environment variable CPX_DEMOD, changed through cpx_demod_set_path()
// 0 auto; 1 "plain": R exponentials per axis, no progression; 2 "libm": the progression with the library's exp / log instead of the
// table-driven ones (the round-4 / early round-5 kernel: kept as the row the table-driven kernel is tested against)
std::atomic<int> g_demod_plain{-1};
int parse_demod_mode(const char *e) { return (e && strcmp(e, "plain") == 0) ? 1 : (e && strcmp(e, "libm") == 0) ? 2 : 0; }
int demod_mode() {
    int v = g_demod_plain.load(std::memory_order_relaxed);
    if (v < 0) {
        v = parse_demod_mode(getenv("CPX_DEMOD"));
        g_demod_plain.store(v, std::memory_order_relaxed);
    }
    return v;
}
bool demod_plain() { return demod_mode() == 1; }

unsigned grid_for(int64_t Ns) {
    int64_t blocks = (Ns + DEMOD_BLOCK - 1) / DEMOD_BLOCK;
    const int64_t cap = 256 * 16;   // 256 CUs x 16 resident blocks; grid-stride beyond that
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

}  // namespace

extern "C" {

int cpx_modem_create(const double *constellation_re_im, int M, cpx_modem **out) {
    CPX_REQUIRE(constellation_re_im && out, CPX_EINVAL, "cpx_modem_create: null pointer");
    CPX_REQUIRE(M >= 2 && (M & (M - 1)) == 0, CPX_EINVAL, "Constellation length must be a power of 2.");
    CPX_REQUIRE(M <= 65536, CPX_ELIMIT, "cpx_modem_create: constellations above 65536 points (16 bits per symbol) are not supported");
    int rc = ensure_device();
    if (rc) return rc;
    cpx_modem *m = new cpx_modem;
    m->M = M;
    m->nbits = 0;
    while ((1 << m->nbits) < M) m->nbits++;
    (void)hipGetDevice(&m->device);
    CPX_HIP(hipMalloc((void **)&m->d_const, sizeof(double) * 2 * M));
    CPX_HIP(hipMemcpy(m->d_const, constellation_re_im, sizeof(double) * 2 * M, hipMemcpyHostToDevice));
    // axis-separable?  label = (a << nh) | b, real part a function of a only, imaginary part of b only (exactly)
    if (m->nbits % 2 == 0 && m->nbits >= 2 && m->nbits <= 8) {
        const int nh = m->nbits / 2, R = 1 << nh;
        std::vector<double> axes(2 * R);
        bool sep = true;
        for (int a = 0; a < R; a++) axes[a] = constellation_re_im[2 * (a << nh)];
        for (int b = 0; b < R; b++) axes[R + b] = constellation_re_im[2 * b + 1];
        for (int mm = 0; mm < M && sep; mm++)
            sep = constellation_re_im[2 * mm] == axes[mm >> nh] && constellation_re_im[2 * mm + 1] == axes[R + (mm & (R - 1))] &&
                  std::isfinite(constellation_re_im[2 * mm]) && std::isfinite(constellation_re_im[2 * mm + 1]);
        for (int a = 0; a < R && sep; a++)                      // distinct grid lines (a degenerate grid keeps the generic path)
            for (int b = a + 1; b < R; b++)
                if (axes[a] == axes[b] || axes[R + a] == axes[R + b]) sep = false;
        if (sep) {
            CPX_HIP(hipMalloc((void **)&m->d_axes, sizeof(double) * 2 * R));
            CPX_HIP(hipMemcpy(m->d_axes, axes.data(), sizeof(double) * 2 * R, hipMemcpyHostToDevice));
            m->separable = true;
            // equally spaced levels in reflected Gray order (what QAMModem builds)?  level j = p0 + j d has label j ^ (j >> 1)
            bool gp = R >= 2;
            for (int ax = 0; ax < 2 && gp; ax++) {
                const double *a = axes.data() + ax * R;
                // (a scaled table, levels * 1/sqrt(Es), sits within a few ulp of the ideal grid: the progression then evaluates
                //  e^{-(x - p)^2 / N0} for a level some 1e-15 |p| off its stored value -- far inside the kernel's own rounding)
                const int top = (R - 1) ^ ((R - 1) >> 1);
                const double p0 = a[0], d = (a[top] - a[0]) / (R - 1);
                double big = 0.0;
                for (int j = 0; j < R; j++) big = std::max(big, std::fabs(a[j]));
                for (int j = 0; j < R && gp; j++) gp = std::fabs(a[j ^ (j >> 1)] - (p0 + j * d)) <= 16.0 * 2.2e-16 * big;
                gp = gp && d != 0.0 && std::isfinite(d);
                m->gp_step[ax] = d;
            }
            m->gp = gp;
        }
    }
    *out = m;
    return CPX_OK;
}

int cpx_demod_set_path(const char *mode) {
    if (mode && mode[0] && strcmp(mode, "auto") != 0 && strcmp(mode, "plain") != 0 && strcmp(mode, "libm") != 0) {
        set_error("cpx_demod_set_path: unknown mode '%s' (auto | plain | libm)", mode);
        return CPX_EINVAL;
    }
    g_demod_plain.store(parse_demod_mode(mode), std::memory_order_relaxed);
    return CPX_OK;
}

int cpx_modem_destroy(cpx_modem *m) {
    if (!m) return CPX_OK;
    (void)hipFree(m->d_const);
    if (m->d_axes) (void)hipFree(m->d_axes);
    delete m;
    return CPX_OK;
}

int cpx_demod_soft_dev(const cpx_modem *m, const double *d_y, int64_t Ns, double noise_var, double *d_llr, void *stream) {
    return cpx_demod_soft_scaled_dev(m, d_y, Ns, noise_var, 1.0, d_llr, stream);
}

int cpx_demod_soft_scaled_dev(const cpx_modem *m, const double *d_y, int64_t Ns, double noise_var, double scale, double *d_llr,
                              void *stream) {
    CPX_TRACE("cpx_demod_soft_scaled_dev");
    CPX_REQUIRE(m, CPX_EINVAL, "demod: null modem");
    if (int rcd = check_handle_device(m->device, "demod")) return rcd;
    CPX_REQUIRE(Ns >= 0, CPX_EINVAL, "demod: negative size");
    if (Ns == 0) return CPX_OK;
    hipStream_t st = pick_stream(stream);
    const double2 *y = reinterpret_cast<const double2 *>(d_y);
    const double2 *c = reinterpret_cast<const double2 *>(m->d_const);
    dim3 grid(grid_for(Ns)), block(DEMOD_BLOCK);
    const bool rcp = noise_var > 1e-290 && noise_var < 1e290;     // 1 / noise_var is a normal number
    if (m->M > MAX_M) {                                           // table too large for the LDS kernels
        switch (m->nbits) {
#define CASE(NB) case NB: hipLaunchKernelGGL(demod_soft_any_kernel<NB>, grid, block, 0, st, y, Ns, c, m->M, noise_var, scale, d_llr); break;
            CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16)
#undef CASE
            default: set_error("demod: unsupported bits per symbol %d", m->nbits); return CPX_ELIMIT;
        }
        CPX_HIP(hipGetLastError());
        note_kernel("demod_soft_any_kernel<%d> (%d points, division form)", m->nbits, m->M);
        return CPX_OK;
    }
    if (precision_fast() && noise_var > 1e-30 && noise_var < 1e30 && std::isfinite(scale)) {   // float32 log-sum-exp variants
        const unsigned rows = (unsigned)std::min<int64_t>((Ns + 63) / 64, (int64_t)256 * 16 * (DEMOD_BLOCK / 64));
        dim3 gridf((rows + DEMOD_BLOCK / 64 - 1) / (DEMOD_BLOCK / 64));
        if (m->separable) {
            switch (m->nbits / 2) {
#define CASE(NH) case NH: hipLaunchKernelGGL(demod_soft_sep_f32_kernel<NH>, gridf, block, 0, st, y, Ns, m->d_axes, noise_var, scale, d_llr); break;
                CASE(1) CASE(2) CASE(3) CASE(4)
#undef CASE
                default: set_error("demod: unsupported bits per symbol %d", m->nbits); return CPX_ELIMIT;
            }
            CPX_HIP(hipGetLastError());
            note_kernel("demod_soft_sep_f32_kernel<%d>", m->nbits / 2);
        } else {
            switch (m->nbits) {
#define CASE(NB) case NB: hipLaunchKernelGGL(demod_soft_f32_kernel<NB>, gridf, block, 0, st, y, Ns, c, m->M, noise_var, scale, d_llr); break;
                CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
                default: set_error("demod: unsupported bits per symbol %d", m->nbits); return CPX_ELIMIT;
            }
            CPX_HIP(hipGetLastError());
            note_kernel("demod_soft_f32_kernel<%d>", m->nbits);
        }
        return CPX_OK;
    }
    const bool al16 = ((uintptr_t)d_llr & 15) == 0;                // the fast kernels store 16 bytes per lane
    const bool gen = demod_mode() != 2 && al16;                   // generic constellations: the table-driven kernel unless "libm"
    const bool gp = m->gp && (m->nbits >= 6 || m->nbits == 2) && !demod_plain();
    const bool tab = gp && m->nbits >= 6 && demod_mode() != 2;       // table-driven exp / log (round 5); "libm" keeps the library's
    if (m->separable && al16) {
        switch (m->nbits / 2) {
#define LAUNCH(NH, RC, GPV) hipLaunchKernelGGL((demod_soft_sep_kernel<NH, RC, GPV>), grid, block, 0, st, y, Ns, m->d_axes, noise_var, \
                                               scale, m->gp_step[0], m->gp_step[1], d_llr)
#define CASE(NH) case NH:                                         \
        if (rcp) LAUNCH(NH, true, false); else LAUNCH(NH, false, false);  \
        break;
#define CASE_GP(NH) case NH:                                      \
        if (gp) { if (rcp) LAUNCH(NH, true, true); else LAUNCH(NH, false, true); }       \
        else { if (rcp) LAUNCH(NH, true, false); else LAUNCH(NH, false, false); }        \
        break;
#define LAUNCH_T(NH, RC) hipLaunchKernelGGL((demod_soft_sep_kernel<NH, RC, true, true>), grid, block, 0, st, y, Ns, m->d_axes, noise_var, \
                                            scale, m->gp_step[0], m->gp_step[1], d_llr)
#define CASE_GT(NH) case NH:                                      \
        if (tab) { if (rcp) LAUNCH_T(NH, true); else LAUNCH_T(NH, false); }              \
        else if (gp) { if (rcp) LAUNCH(NH, true, true); else LAUNCH(NH, false, true); }  \
        else { if (rcp) LAUNCH(NH, true, false); else LAUNCH(NH, false, false); }        \
        break;
            CASE_GP(1) CASE(2) CASE_GT(3) CASE_GT(4)              // two levels: closed form; 8 and 16 levels: progression (4 exp instead of R)
#undef CASE
#undef CASE_GP
#undef CASE_GT
#undef LAUNCH_T
#undef LAUNCH
            default: set_error("demod: unsupported bits per symbol %d", m->nbits); return CPX_ELIMIT;
        }
    } else {
        switch (m->nbits) {
#define CASE(NB) case NB:                                                                                                \
        if (gen && rcp) hipLaunchKernelGGL((demod_soft_gen_kernel<NB, true>), grid, block, 0, st, y, Ns, c, noise_var, scale, d_llr);   \
        else if (gen) hipLaunchKernelGGL((demod_soft_gen_kernel<NB, false>), grid, block, 0, st, y, Ns, c, noise_var, scale, d_llr);   \
        else if (rcp) hipLaunchKernelGGL((demod_soft_kernel<NB, true>), grid, block, 0, st, y, Ns, c, m->M, noise_var, scale, d_llr);   \
        else hipLaunchKernelGGL((demod_soft_kernel<NB, false>), grid, block, 0, st, y, Ns, c, m->M, noise_var, scale, d_llr);   \
        break;
            CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
            default: set_error("demod: unsupported bits per symbol %d", m->nbits); return CPX_ELIMIT;
        }
    }
    CPX_HIP(hipGetLastError());
    if (m->separable && al16) note_kernel("demod_soft_sep_kernel<%d,%s%s%s>", m->nbits / 2, rcp ? "rcp" : "div", gp ? ",gp" : "", tab ? ",tab" : "");
    else note_kernel("%s<%d,%s>", gen ? "demod_soft_gen_kernel" : "demod_soft_kernel", m->nbits, rcp ? "rcp" : "div");
    return CPX_OK;
}

int cpx_demod_hard_dev(const cpx_modem *m, const double *d_y, int64_t Ns, int8_t *d_bits, void *stream) {
    CPX_TRACE("cpx_demod_hard_dev");
    CPX_REQUIRE(m, CPX_EINVAL, "demod: null modem");
    if (int rcd = check_handle_device(m->device, "demod")) return rcd;
    CPX_REQUIRE(Ns >= 0, CPX_EINVAL, "demod: negative size");
    if (Ns == 0) return CPX_OK;
    hipStream_t st = pick_stream(stream);
    dim3 grid(grid_for(Ns)), block(DEMOD_BLOCK);
    const double2 *y = reinterpret_cast<const double2 *>(d_y);
    const double2 *c = reinterpret_cast<const double2 *>(m->d_const);
    if (m->M > MAX_M) {
        hipLaunchKernelGGL(demod_hard_any_kernel, grid, block, 0, st, y, Ns, c, m->M, m->nbits, d_bits);
        CPX_HIP(hipGetLastError());
        note_kernel("demod_hard_any_kernel (%d points)", m->M);
        return CPX_OK;
    }
    if (m->separable) {
        switch (m->nbits / 2) {
#define CASE(NH) case NH: hipLaunchKernelGGL(demod_hard_sep_kernel<NH>, grid, block, 0, st, y, Ns, c, m->d_axes, d_bits); break;
            CASE(1) CASE(2) CASE(3) CASE(4)
#undef CASE
            default: set_error("demod: unsupported bits per symbol %d", m->nbits); return CPX_ELIMIT;
        }
    } else {
        hipLaunchKernelGGL(demod_hard_kernel, grid, block, 0, st, y, Ns, c, m->M, m->nbits, d_bits);
    }
    CPX_HIP(hipGetLastError());
    if (m->separable) note_kernel("demod_hard_sep_kernel<%d>", m->nbits / 2);
    else note_kernel("demod_hard_kernel");
    return CPX_OK;
}

int cpx_demod_soft(const cpx_modem *m, const double *y_re_im, int64_t Ns, double noise_var, double *llr) {
    CPX_TRACE("cpx_demod_soft");
    CPX_REQUIRE(m && (y_re_im || Ns == 0) && (llr || Ns == 0), CPX_EINVAL, "demod: null pointer");
    int rc = ensure_device();
    if (rc) return rc;
    if (Ns == 0) return CPX_OK;
    DevBuf din, dout;
    const size_t out_bytes = sizeof(double) * (size_t)Ns * m->nbits;
    if ((rc = din.alloc(sizeof(double) * 2 * (size_t)Ns))) return rc;
    if ((rc = dout.alloc(out_bytes))) return rc;
    hipStream_t st = lib_stream();
    CPX_HIP(hipMemcpyAsync(din.p, y_re_im, sizeof(double) * 2 * (size_t)Ns, hipMemcpyHostToDevice, st));
    if ((rc = cpx_demod_soft_dev(m, din.as<double>(), Ns, noise_var, dout.as<double>(), st))) return rc;
    if ((rc = d2h_pageable(llr, dout.p, out_bytes, st))) return rc;
    CPX_HIP(hipStreamSynchronize(st));
    return CPX_OK;
}

int cpx_demod_hard(const cpx_modem *m, const double *y_re_im, int64_t Ns, int8_t *bits) {
    CPX_TRACE("cpx_demod_hard");
    CPX_REQUIRE(m && (y_re_im || Ns == 0) && (bits || Ns == 0), CPX_EINVAL, "demod: null pointer");
    int rc = ensure_device();
    if (rc) return rc;
    if (Ns == 0) return CPX_OK;
    DevBuf din, dout;
    const size_t out_bytes = (size_t)Ns * m->nbits;
    if ((rc = din.alloc(sizeof(double) * 2 * (size_t)Ns))) return rc;
    if ((rc = dout.alloc(out_bytes))) return rc;
    hipStream_t st = lib_stream();
    CPX_HIP(hipMemcpyAsync(din.p, y_re_im, sizeof(double) * 2 * (size_t)Ns, hipMemcpyHostToDevice, st));
    if ((rc = cpx_demod_hard_dev(m, din.as<double>(), Ns, dout.as<int8_t>(), st))) return rc;
    CPX_HIP(hipMemcpyAsync(bits, dout.p, out_bytes, hipMemcpyDeviceToHost, st));
    CPX_HIP(hipStreamSynchronize(st));
    return CPX_OK;
}

// ---- fused link front end (link_front_kernel) ----------------------------------------------------------------------------------
struct cpx_link_front {
    __attribute__((visibility("hidden"))) ~cpx_link_front() = default;
    const cpx_trellis *t;
    const cpx_modem *m;
    int device;
    int nbits, mem, ntx, nsym, nde, lead;
    int32_t *d_jo = nullptr, *d_pg = nullptr;
};

int cpx_link_front_create(const cpx_trellis *t, const cpx_modem *m, int64_t nbits, const int32_t *keep_idx, int64_t ntx,
                          const int32_t *pos_idx, int64_t nde, cpx_link_front **out) {
    CPX_TRACE("cpx_link_front_create");
    CPX_REQUIRE(t && m && out, CPX_EINVAL, "link_front: null pointer");
    if (int rcd = check_handle_device(t->device, "link_front")) return rcd;
    if (int rcd = check_handle_device(m->device, "link_front")) return rcd;
    CPX_REQUIRE(nbits > 0 && ntx > 0 && nde >= ntx, CPX_EINVAL, "link_front: sizes");
    // what the kernel is built for -- anything else stays on the staged kernels (CPX_ELIMIT tells the caller so)
    int mem = 0;
    while ((1 << mem) < t->S) mem++;
    bool ff = t->k == 1 && t->I == 2 && mem >= 1 && mem <= 6 && (1 << mem) == t->S && t->n >= 1 && t->n <= 8;
    for (int s2 = 0; s2 < t->S && ff; s2++)
        for (int b2 = 0; b2 < 2; b2++)
            if (t->next_state[s2 * 2 + b2] != ((b2 << (mem - 1)) | (s2 >> 1))) ff = false;
    if (!ff) { set_error("link_front: not a feed-forward k = 1 shift-register trellis of <= 64 states"); return CPX_ELIMIT; }
    const int nb = m->nbits;
    if (!m->separable || (nb != 2 && nb != 4 && nb != 6 && nb != 8)) { set_error("link_front: modem is not a square QAM of 4..256 points"); return CPX_ELIMIT; }
    if ((nb >= 6 || nb == 2) && !m->gp) { set_error("link_front: axis levels are not equally spaced Gray levels"); return CPX_ELIMIT; }
    if (ntx % nb) { set_error("link_front: %lld transmitted bits are not a whole number of symbols", (long long)ntx); return CPX_ELIMIT; }
    const int64_t nsym = ntx / nb, ncoded = nbits * t->n;
    if (nbits >= (1 << 24) || nde >= (1 << 26) || nbits * nsym >= (1ll << 31)) { set_error("link_front: frame too long"); return CPX_ELIMIT; }
    // transmitted bit q comes from coded position keep[q] (increasing) and goes to decoder input pos[q] (increasing); the kernel's tables:
    // jo[q] = (trellis step << 4) | shift of that output bit in the code word, pg[q] = pos | (punctured positions behind it << 26)
    std::vector<int32_t> jo((size_t)ntx), pg((size_t)ntx);
    int64_t last_c = -1, last_p = -1;
    for (int64_t q = 0; q < ntx; q++) {
        const int64_t c = keep_idx ? keep_idx[q] : q, pp = pos_idx ? pos_idx[q] : q;
        CPX_REQUIRE(c > last_c && c < ncoded && pp > last_p && pp < nde, CPX_EINVAL, "link_front: index tables must increase and stay in range");
        last_c = c; last_p = pp;
        const int64_t next = q + 1 < ntx ? (pos_idx ? pos_idx[q + 1] : q + 1) : nde;
        if (next - pp - 1 > 63) { set_error("link_front: more than 63 punctured positions in a row"); return CPX_ELIMIT; }
        jo[(size_t)q] = (int32_t)(((c / t->n) << 4) | (t->n - 1 - c % t->n));
        pg[(size_t)q] = (int32_t)((uint32_t)pp | ((uint32_t)(next > pp ? next - pp - 1 : 0) << 26));
    }
    const int64_t lead = pos_idx ? pos_idx[0] : 0;
    // every symbol's message window must fit the kernel's 64-bit register (four Philox values of 16 bits)
    for (int64_t sy = 0; sy < nsym; sy++) {
        const int64_t c0 = keep_idx ? keep_idx[sy * nb] : sy * nb, c1 = keep_idx ? keep_idx[sy * nb + nb - 1] : sy * nb + nb - 1;
        const int64_t lo = std::min<int64_t>(c0 / t->n - mem, sy * nbits / nsym), hi = std::min<int64_t>(std::max<int64_t>(c1 / t->n, (sy + 1) * nbits / nsym - 1), nbits - 1);
        if (hi - lo + 1 > 49) { set_error("link_front: a symbol depends on %lld message bits (limit 49)", (long long)(hi - lo + 1)); return CPX_ELIMIT; }
    }
    cpx_link_front *lf = new (std::nothrow) cpx_link_front();
    CPX_REQUIRE(lf, CPX_ENOMEM, "link_front: out of memory");
    lf->t = t; lf->m = m; lf->device = t->device; lf->nbits = (int)nbits; lf->mem = mem; lf->ntx = (int)ntx; lf->nsym = (int)nsym; lf->nde = (int)nde;
    lf->lead = (int)lead;
    auto up = [&](const std::vector<int32_t> &h, int32_t **d) -> int {
        if (hipMalloc((void **)d, sizeof(int32_t) * h.size()) != hipSuccess) return CPX_ENOMEM;
        return hipMemcpy(*d, h.data(), sizeof(int32_t) * h.size(), hipMemcpyHostToDevice) == hipSuccess ? CPX_OK : CPX_EHIP;
    };
    int rc = up(jo, &lf->d_jo);
    if (!rc) rc = up(pg, &lf->d_pg);
    if (rc) { cpx_link_front_destroy(lf); set_error("link_front: table upload failed"); return rc; }
    *out = lf;
    return CPX_OK;
}

int cpx_link_front_destroy(cpx_link_front *lf) {
    if (!lf) return CPX_OK;
    if (lf->d_jo) (void)hipFree(lf->d_jo);
    if (lf->d_pg) (void)hipFree(lf->d_pg);
    delete lf;
    return CPX_OK;
}

int cpx_link_front_run_dev(const cpx_link_front *lf, int64_t T, double noise_var, double scale_re, double scale_im, double llr_scale,
                           uint64_t seed, uint64_t stream_bits, uint64_t stream_noise, uint8_t *d_msg, double *d_llr, double *d_rx_re_im,
                           void *stream) {
    CPX_TRACE("cpx_link_front_run_dev");
    CPX_REQUIRE(lf && (T == 0 || (d_msg && d_llr)), CPX_EINVAL, "link_front: null pointer");
    if (int rcd = check_handle_device(lf->device, "link_front")) return rcd;
    CPX_REQUIRE(T >= 0, CPX_EINVAL, "link_front: negative size");
    if (T == 0) return CPX_OK;
    if (T * (int64_t)lf->nsym >= (1ll << 31) / 8) { set_error("link_front: %lld transmissions in one call (limit: 2^28 symbols)", (long long)T); return CPX_ELIMIT; }
    const cpx_modem *m = lf->m;
    // the demodulator variant the staged path would run for this modem and mode (cpx_demod_soft_scaled_dev); other modes: staged
    const bool rcp = noise_var > 1e-290 && noise_var < 1e290;
    if (!rcp || precision_fast() || demod_mode() != 0) {
        set_error("link_front: only the default float64 demodulator path is fused (noise_var %g, mode %d)", noise_var, demod_mode());
        return CPX_ELIMIT;
    }
    LinkFrontParams p;
    p.T = (int32_t)T; p.nbits = lf->nbits; p.mem = lf->mem; p.ntx = lf->ntx; p.nsym = lf->nsym; p.nde = lf->nde;
    p.jo = lf->d_jo; p.pg = lf->d_pg; p.lead = lf->lead; p.out_tab = lf->t->d_out; p.axes = m->d_axes;
    p.step_x = m->gp_step[0]; p.step_y = m->gp_step[1]; p.noise_var = noise_var; p.scale_re = scale_re; p.scale_im = scale_im;
    p.llr_scale = llr_scale; p.seed = seed; p.stream_bits = stream_bits; p.stream_noise = stream_noise;
    p.msg = d_msg; p.llr = d_llr; p.rx = reinterpret_cast<double2 *>(d_rx_re_im);
    dim3 grid(grid_for(T * (int64_t)lf->nsym)), block(DEMOD_BLOCK);
    hipStream_t st = pick_stream(stream);
    switch (m->nbits / 2) {                                       // <NH, RCP, GP, TAB> as in cpx_demod_soft_scaled_dev's default mode
        case 1: hipLaunchKernelGGL((link_front_kernel<1, true, true, false>), grid, block, 0, st, p); break;
        case 2: hipLaunchKernelGGL((link_front_kernel<2, true, false, false>), grid, block, 0, st, p); break;
        case 3: hipLaunchKernelGGL((link_front_kernel<3, true, true, true>), grid, block, 0, st, p); break;
        default: hipLaunchKernelGGL((link_front_kernel<4, true, true, true>), grid, block, 0, st, p); break;
    }
    CPX_HIP(hipGetLastError());
    note_kernel("link_front_kernel<%d> (bits, conv_encode, puncture, modulate, AWGN, soft demod, depuncture fused)", m->nbits / 2);
    return CPX_OK;
}

}  // extern "C"
