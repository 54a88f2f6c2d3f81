// Device helpers shared by the LDPC belief-propagation kernels (ldpc.hip: tiled HBM-resident path, ldpc_resident.hip:
// LDS-resident path).  Both paths perform the same float64 operations in the same order per edge.
#pragma once
#include "cpx_internal.h"
#include "cpx_math.h"

namespace cpx {

__device__ __forceinline__ double clip_nan(double v, double lo, double hi) {
    // np.clip propagates NaN
    return (v != v) ? v : fmin(fmax(v, lo), hi);
}

__device__ __forceinline__ double min_f64(double a, double b) {     // one v_min_f64 (fmin adds two canonicalising v_max)
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double max_f64(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double ntload(const double *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void ntstore(double v, double *p) { __builtin_nontemporal_store(v, p); }

// tanh(v/2) and 2*atanh(x) for the sum-product pass.  ocml's tanh/atanh are correctly rounded to < 1 ulp through
// double-double arithmetic (~300 VALU instructions per edge for the pair, which bounded the check pass); these
// evaluate the same functions through one exp / one log with an ABSOLUTE error of a few 1e-16:
//   tanh(v/2) = sign(v) (1 - e) / (1 + e),  e = exp(-|v|)        (cancellation in 1 - e only costs relative accuracy
//                                                                  of results that are themselves ~|v|/2 << 1)
//   2 atanh(x) = log((1 + x) / (1 - x))                            (1 - x is exact for x > 1/2; x = +-1 gives +-inf)
// The decoder's sensitivity to the last ulp of these functions is the same either way ("LDPC-SPA note").
// Round 5: near saturation the quotient form loses to a real tanh -- (1 - e) and (1 + e) are each rounded before the division, up to
// 1.5 ulp of a result whose distance from 1 is what 2 atanh amplifies -- and that, not the libm, was why the engine's exact-order
// row missed the reference on TWICE as many saturated values as the glibc oracle (1.70 % against 0.90 % in [50, 100), 13.6 % against
// 6.7 % above 100 on the 72 live-reference blocks; reproduced on the CPU by putting this very formula into the oracle, and undone by
// the form below: profiles/r05_spa_tolerance.md).  fdlibm's own large-argument form, tanh(x) = 1 - 2 / (expm1(2x) + 2), rewritten in
// e = exp(-|v|): 1 - 2e / (1 + e) -- the small term carries the roundings, the result is within ~0.5 ulp from e <= 1/8 on.
__device__ __forceinline__ double tanh_from_e(double e) {
    return e <= 0.125 ? 1.0 - (2.0 * e) / (1.0 + e) : (1.0 - e) / (1.0 + e);
}
__device__ __forceinline__ double tanh_half(double v) {
    const double e = exp(-fabs(v));
    return __builtin_copysign(tanh_from_e(e), v);                 // NaN propagates through exp (e <= 0.125 is false for a NaN)
}
__device__ __forceinline__ double atanh_twice(double x) { return fast_log((1.0 + x) / (1.0 - x)); }

// ---- sum-product check node with ONE division per edge, exact redo near saturation ----------------------------------------
// The reference's row (:209-227) costs three float64 divisions per edge in this engine's exact-order form: tanh(m/2) as
// (1 - e) / (1 + e), the reciprocal 1 / t, and (1 + x) / (1 - x) inside 2 atanh(x).  Algebraically, with e_j = exp(-|m_j|),
// u_j = sign(m_j) (1 - e_j), w_j = 1 + e_j, U = prod u_j, W = prod w_j:
//     x_j = P / t_j = (U / W) (w_j / u_j)     and     (1 + x_j) / (1 - x_j) = (W u_j + U w_j) / (W u_j - U w_j)
// -- one division per edge (measured in round 2: 8.1 instead of 9.6 ms for config 4's share at 3 dB, identical dec_word and
// iteration counts, <= 3e-7 from the exact-order row up to |LLR| = 26).  It was not shipped then because of ONE regime: where
// |x| is within a few ulp of 1 the reference's result is decided by how ITS sequence fl(fl(1 / t) P) rounds -- clip to 1 gives
// atanh = inf -> 500, one ulp below gives 37.4 -- and the rearranged row rounds differently there.  Now a row is first tested
// for that regime, cheaply and conservatively: the largest |x_j| of a row belongs to the edge with the smallest |m_j|, so
//     near  <=>  not ( |U| w* < |W| |u*| (1 - 2^-32) ),      (u*, w*) of that edge; also true for NaN / inf / t = 0
// and a near row is evaluated by the exact-order sequence (spa_exact_t + the caller's second loop), bit for bit what the engine
// did for every row before.  Rows that are not near have |x_j| < 1 - 2^-32 for every edge: messages below 22.9, no clip binds.
// Why 2^-32 and not "a few ulp": 2 atanh(x) turns a relative error eps of x into eps / (1 - |x|) of the message.  The
// exact-order sequence shares its roundings with the reference's (the two agree far better than that bound); the rearranged
// row does not, its eps ~ 1e-15 is independent: 1 - |x| >= 2^-32 keeps the difference below 4e-6, inside the 1e-5 contract
// (at 2^-44 it reached 1e-2 on messages of 26 - 31).
__device__ __forceinline__ void spa_in(double m, double &se, double &u, double &w) {
    const double e = exp(-fabs(m));
    se = __builtin_copysign(e, m);                                // what the row keeps per edge: e with the sign of m
    u = __builtin_copysign(1.0 - e, m);
    w = 1.0 + e;
}
__device__ __forceinline__ bool spa_row_near(double U, double W, double emax) {
    return !(fabs(U) * (1.0 + emax) < fabs(W) * (1.0 - emax) * (1.0 - 0x1p-32));
}
__device__ __forceinline__ double spa_out_fast(double U, double W, double se) {
    const double e = fabs(se);
    const double n1 = W * __builtin_copysign(1.0 - e, se), n2 = U * (1.0 + e);
    // 2 atanh(x), |x| < 1 - 2^-32: no clip binds.  Round 4: a row that is not "near" has |n1 - n2| >= 2^-32 |n1| with n1 = W u_j a
    // product of normal numbers (|u_j| >= 2^-53, W in [1, 2^deg]) and a quotient in (2^-33, 2^33): both divisions -- this one and
    // the one inside the logarithm -- run as reciprocal + two Newton steps + residual (cpx_math.h div_nr: 8 instead of 12
    // instructions, < 2 ulp) and the logarithm drops its special-value selects; the row stays inside its 4e-6 budget.
#ifdef CPX_SPA_IEEE_DIV                                           /* A/B builds (scripts/build_variant.sh): the round-3 row */
    return fast_log((n1 + n2) / (n1 - n2));
#else
    return fast_log<false, true>(div_nr(n1 + n2, n1 - n2));
#endif
}
// NaN signs.  An LLR of exactly 0 is a case the reference expects (ldpc.py:214: "Runtime Warnings are expected when llr = 0"):
// tanh(0) = 0, 1 / 0 = inf, inf * 0 = NaN, and from there the block fills with NaN -- but dec_word = signbit(out_llrs) (:193,
// :248) and the early-termination test (:205) read the SIGN of those NaNs.  What NumPy on x86 does, and this path reproduces
// (tests/golden/abnormal.npz, spaz_*): an invalid operation GENERATES a negative NaN (x86's default, 0xFFF8...; gfx950's is
// positive); np.tanh and the complex log2 / exp2 product return a POSITIVE NaN for a NaN argument whatever its sign; every other
// operation of the loop passes a NaN on with its sign.  A NaN never takes the fast row (spa_row_near sees U or W).
__device__ __forceinline__ double spa_exact_t(double se) {        // == tanh_half(m) of the edge, from its stored e
    const double e = fabs(se);
    const double t = __builtin_copysign(tanh_from_e(e), se);
    return se == se ? t : __builtin_nan("");                      // np.tanh(NaN) is +NaN
}
__device__ __forceinline__ double spa_out_exact(double t, double prod) {
    double x = (1.0 / t) * prod;                                  // data = 1/data; multiply(msg_products) (:222-223)
    // generated here (inf * 0): negative.  Propagated: t and prod come out of tanh / the product, i.e. positive.  atanh, * 2 and
    // the clips (:224-227) pass either on.
    if (x != x) return (t == t && prod == prod) ? -__builtin_nan("") : __builtin_nan("");
    x = clip_nan(x, -1.0, 1.0);                                   // (:224)
    x = atanh_twice(x);                                           // (:225-226)
    return clip_nan(x, -500.0, 500.0);                            // (:227)
}

// ---- min-sum (:229-238): the messages of a row are (+-) one of TWO magnitudes, so the row is kept as a record
//   rec[0] = min1 = smallest |v->c message| of the row, rec[1] = min2 = second smallest (ties: a later equal
//   value), rec[2] = meta: bits 0..7 position of min1, bit 8 parity of the negatives, bits 32..63 negative mask.
// Edge j receives  (-1)^(negatives among the others) * (j == imin ? min2 : min1)  -- exactly
// sign(other).prod() * abs(other).min(): a zero among the others makes the minimum zero by itself.
struct MsaRec { double m1, m2; unsigned neg; int imin, par; };

__device__ __forceinline__ MsaRec msa_load(const double *__restrict__ rec) {
    MsaRec r;
    r.m1 = ntload(&rec[0]);
    r.m2 = ntload(&rec[64]);
    const double meta = ntload(&rec[128]);
    const int lo = __double2loint(meta);
    r.neg = (unsigned)__double2hiint(meta);
    r.imin = lo & 0xff;
    r.par = (lo >> 8) & 1;
    return r;
}
// the message edge j of the row received, negated or not (flip = 1 returns -R)
__device__ __forceinline__ double msa_edge(const MsaRec &r, int j, int flip) {
    const double mn = (j == r.imin) ? r.m2 : r.m1;
    const unsigned ng = ((r.neg >> j) ^ (unsigned)r.par ^ (unsigned)flip) & 1u;
    return __hiloint2double(__double2hiint(mn) | (int)(ng << 31), __double2loint(mn));
}

}  // namespace cpx
