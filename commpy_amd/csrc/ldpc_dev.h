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
__device__ __forceinline__ double ntload(const double *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void ntstore(double v, double *p) { __builtin_nontemporal_store(v, p); }

// tanh(v/2) and 2*atanh(x) for the sum-product pass.  ocml's tanh/atanh are correctly rounded to < 1 ulp through
// double-double arithmetic (~300 VALU instructions per edge for the pair, which bounded the check pass); these
// evaluate the same functions through one exp / one log with an ABSOLUTE error of a few 1e-16:
//   tanh(v/2) = sign(v) (1 - e) / (1 + e),  e = exp(-|v|)        (cancellation in 1 - e only costs relative accuracy
//                                                                  of results that are themselves ~|v|/2 << 1)
//   2 atanh(x) = log((1 + x) / (1 - x))                            (1 - x is exact for x > 1/2; x = +-1 gives +-inf)
// The decoder's sensitivity to the last ulp of these functions is the same either way ("LDPC-SPA note").
__device__ __forceinline__ double tanh_half(double v) {
    const double e = exp(-fabs(v));
    const double t = (1.0 - e) / (1.0 + e);
    return __builtin_copysign(t, v);                              // NaN propagates through exp
}
__device__ __forceinline__ double atanh_twice(double x) { return fast_log((1.0 + x) / (1.0 - x)); }

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
