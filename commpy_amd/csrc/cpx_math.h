// Device math shared by the kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace cpx {

// Natural logarithm, float64, error < 1 ulp: the classic argument reduction x = 2^k * m, m in [sqrt(1/2), sqrt(2)),
// s = f / (2 + f) with f = m - 1, and a degree-14 odd series in s evaluated as two interleaved polynomials in s^4
// (the construction of Sun's fdlibm e_log.c, coefficients from a Remez fit on [0, 0.1716]).  About 45 VALU
// instructions on gfx950; ocml's log (double-double, 98 instructions) bounded the soft demodulator, the
// sum-product check pass and the branch-metric stage.  Special values as libm: log(0) = -inf, log(x < 0) = NaN,
// log(inf) = inf, a NaN argument is returned unchanged (its sign decides dec_word in
// the LDPC passes exactly as it does through libm); denormals are handled by v_frexp.
// SPECIAL = false drops the four special-value selects: for arguments known to be finite and >= 1 (the Viterbi branch
// metrics take log(exp(r) + 1) with |r| <= 500) the result is bit-identical and 4 compares + 8 selects shorter.
// x / y for NORMAL, well-scaled operands (no denormals, no overflow of the quotient, y != 0): hardware reciprocal, one Newton
// step, one residual correction -- 6 instructions against the 12 of the IEEE division sequence (v_div_scale x 2, v_rcp, five
// v_fma, v_div_fmas, v_div_fixup).  Only where a caller's contract does not need the IEEE sequence's guarantees and its operands
// are known to be in range: the sum-product rows (ldpc_dev.h, 4e-6 budget), the soft demodulator's quotients.
__device__ __forceinline__ double div_nr(double x, double y) {
    // v_rcp_f64 is good to ~2^-25 (measured, scripts/micro/div_nr_check.hip): ONE Newton step brings the reciprocal to ~2^-50 and the
    // residual step then returns the correctly rounded quotient on every one of 5e8 random operand pairs -- exactly what the form with
    // two Newton steps (rounds 3 / 4a, 8 instructions) returned, with 6 instructions
    double r = __builtin_amdgcn_rcp(y);
    const double e = __builtin_fma(-y, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q = x * r;
    return __builtin_fma(__builtin_fma(-y, q, x), r, q);
}

// No Newton step: v_rcp_f64 + residual (4 instructions): up to 10 ulp off (same measurement).  Only where the quotient's error budget is
// absolute and generous: e = exp(-|m|) of the ratio-domain sum-product row (2e-15 on m).
__device__ __forceinline__ double div_nr0(double x, double y) {
    const double r = __builtin_amdgcn_rcp(y);
    const double q = x * r;
    return __builtin_fma(__builtin_fma(-y, q, x), r, q);
}

// NR = true: the division of the argument reduction by div_nr (its operands are in [-0.3, 0.42] / [1.7, 2.42]): < 2 ulp instead of
// < 1 ulp -- NOT for the Viterbi branch metrics or anything else that is compared bit for bit.
template <bool SPECIAL = true, bool NR = false>
__device__ __forceinline__ double fast_log(double x) {
    constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    constexpr double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                     Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                     Lg7 = 1.479819860511658591e-01;
    double m = __builtin_amdgcn_frexp_mant(x);                    // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool lo = m < 7.07106781186547524401e-01;
    m = lo ? m + m : m;                                           // [sqrt(1/2), sqrt(2))
    e = lo ? e - 1 : e;
    const double f = m - 1.0;
    const double s = NR ? div_nr(f, 2.0 + f) : f / (2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * __builtin_fma(w, __builtin_fma(w, Lg6, Lg4), Lg2);
    const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    double r = dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    if (SPECIAL) {
        r = (x == 0.0) ? -__builtin_huge_val() : r;
        r = (x == __builtin_huge_val()) ? x : r;
        r = (x < 0.0) ? __builtin_nan("") : r;
        r = (x != x) ? x : r;                                     // a NaN argument is returned as is (sign and payload)
    }
    return r;
}

}  // namespace cpx
