"""Where does map_decode stop agreeing with the reference?  Sweeps symbol amplitude, noise variance and the scale of L_int far beyond
the operating range and reports, per regime, finiteness / NaN pattern mismatches, the largest relative deviation on values both sides
hold finite, and decision mismatches ("BCJR outside the operating range")."""
import sys, numpy as np, warnings
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import oracle
from helpers import make_trellis
from commpy_amd.channelcoding import map_decode
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    trs = [make_trellis("rsc_legacy_4"), make_trellis("rsc_legacy_8")]
rs = np.random.RandomState(0)
worst = {}
for case in range(400):
    tr = trs[case % 2]
    N = int(rs.randint(5, 120)); B = 4
    amp = float(rs.choice([1.0, 5.0, 20.0])); nv = float(rs.choice([0.02, 0.1, 1.0])); lsc = float(rs.choice([0.0, 5.0, 60.0]))
    s_ = (rs.choice([-1.0, 1.0], size=(B, N)) + rs.randn(B, N) * 0.5) * amp
    p_ = (rs.choice([-1.0, 1.0], size=(B, N)) + rs.randn(B, N) * 0.5) * amp
    L = rs.randn(B, N) * lsc
    Le, bits = map_decode(s_, p_, tr, nv, L, "decode")
    for b in range(B):
        Lo, bo = oracle.map_decode(s_[b], p_[b], tr, nv, L[b], "decode")
        fe, fo = np.isfinite(Le[b]), np.isfinite(Lo)
        key = (amp, nv, lsc)
        w = worst.setdefault(key, [0, 0.0, 0, 0])
        w[0] += int(np.sum(fe != fo)) + int(np.sum(np.isnan(Le[b]) != np.isnan(Lo)))
        both = fe & fo
        if both.any():
            d = np.abs(Le[b][both] - Lo[both]); m = np.abs(Lo[both])
            w[1] = max(w[1], float(np.max(d / np.maximum(1.0, m))))
        w[2] += int(np.sum((bits[b] != bo) & fo & (np.abs(Lo) > 1e-5)))
        w[3] += N
for k in sorted(worst):
    print("amp %5.1f nv %5.2f Lscale %5.1f: finiteness/NaN pattern mismatches %4d, max rel dev %.2e, bit mismatches %d  (of %d values)" % (k + tuple(worst[k])), flush=True)
