// Hard-decision demodulation of ONE symbol, shared by demod.hip (element-wise kernels) and viterbi.hip (the fused
// hard-demod -> hard-Viterbi entry point): Modem.demodulate(y, 'hard'), /root/reference/commpy/modulation.py:121-123.
#pragma once
#include <hip/hip_runtime.h>

namespace cpx {

// abs(y - c[:, None]).argmin(0): first minimum of |y - c_m| over the constellation, |.| = hypot like np.abs (:122)
__device__ __forceinline__ int hard_scan(const double2 *c, int M, double2 cur) {
    int best = 0;
    double bd = hypot(cur.x - c[0].x, cur.y - c[0].y);
    for (int m = 1; m < M; m++) {
        const double a = hypot(cur.x - c[m].x, cur.y - c[m].y);
        if (a < bd) { bd = a; best = m; }
    }
    return best;
}

// Axis-separable square constellations (what QAMModem builds): label = (a << NH) | b, point = xs[a] + 1j*ys[b],
// axes = [xs | ys].  The nearest grid line per axis gives the nearest point; a symbol whose two best distances on an axis
// are closer than 1e-12 (relative) is re-decided with the reference's full hypot scan, so the first-minimum rule of
// argmin holds on decision boundaries too.
template <int NH>
__device__ __forceinline__ int hard_sep(const double2 *c, const double *axes, double2 cur) {
    constexpr int R = 1 << NH;
    int ia = 0, ib = 0;
    double da = fabs(cur.x - axes[0]), db = fabs(cur.y - axes[R]), da2 = __builtin_huge_val(), db2 = da2;
#pragma unroll
    for (int a = 1; a < R; a++) {
        const double dx = fabs(cur.x - axes[a]), dy = fabs(cur.y - axes[R + a]);
        if (dx < da) { da2 = da; da = dx; ia = a; } else if (dx < da2) da2 = dx;
        if (dy < db) { db2 = db; db = dy; ib = a; } else if (dy < db2) db2 = dy;
    }
    int best = (ia << NH) | ib;
    if (da2 - da <= 1e-12 * da2 || db2 - db <= 1e-12 * db2) best = hard_scan(c, R * R, cur);
    return best;
}

// run-time NH (1..4) or 0 = generic constellation
__device__ __forceinline__ int hard_label(const double2 *c, const double *axes, int M, int nh, double2 cur) {
    switch (nh) {
        case 1: return hard_sep<1>(c, axes, cur);
        case 2: return hard_sep<2>(c, axes, cur);
        case 3: return hard_sep<3>(c, axes, cur);
        case 4: return hard_sep<4>(c, axes, cur);
        default: return hard_scan(c, M, cur);
    }
}

}  // namespace cpx
