import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle
from helpers import ldpc_params
from commpy_amd import _lib
from test_ldpc_resident_gpu import _decode, _staggered

def report(tag, a, b):
    d1, o1, i1 = a; d2, o2, i2 = b
    with np.errstate(invalid="ignore"):
        fin = np.isfinite(o1) & np.isfinite(o2)
        dev = np.abs(o1 - o2)
        mag = np.abs(o1)
        nb = o1.shape[1]
        badblk = [int(b_) for b_ in range(nb) if i1[b_] != i2[b_] or not np.array_equal(d1[:, b_], d2[:, b_])]
        print(tag, "its equal", np.array_equal(i1, i2), "dec equal", np.array_equal(d1, d2), "nan pattern equal", np.array_equal(np.isnan(o1), np.isnan(o2)),
              "bad blocks", badblk[:10], len(badblk), "of", nb)
        for lo, hi in ((0, 10), (10, 26), (26, 1e9)):
            m = fin & (mag >= lo) & (mag < hi)
            if m.any():
                print("   |LLR| in [%g,%g): n=%d max dev %.3e, frac <=1e-5: %.6f, max rel %.3e" % (lo, hi, m.sum(), dev[m].max(), np.mean(dev[m] <= 1e-5), (dev[m] / np.maximum(mag[m], 1)).max()))

p = ldpc_params("n1944")
for alg, iters in (("SPA", 14), ("SPA", 2)):
    rs = np.random.RandomState(5)
    llr = _staggered(rs, 2100, 1944, 2.0 / 3, [0.5, 2.0, 2.6, 3.2, 4.5, 30.0])
    r = {}
    for path in ("tiled", "resident", "resident-log"):
        d, o, i, x, k = _decode(_lib, path, llr, p, alg, iters)
        r[path] = (d, o, i)
    report("n1944 %s %d ratio vs tiled" % (alg, iters), r["tiled"], r["resident"])
    print("   log vs tiled identical:", all(np.array_equal(u, v, equal_nan=True) for u, v in zip(r["tiled"], r["resident-log"])))
for name, n in (("gallager96", 96), ("wimax1440", 1440)):
    p2 = ldpc_params(name)
    rs = np.random.RandomState(11)
    llr = _staggered(rs, 777, n, 0.5, [1.0, 2.5, 4.0, 30.0])
    r = {}
    for path in ("tiled", "resident", "resident-log"):
        d, o, i, x, k = _decode(_lib, path, llr, p2, "SPA", 8)
        r[path] = (d, o, i)
    report(name + " ratio vs tiled", r["tiled"], r["resident"])
    print("   log vs tiled identical:", all(np.array_equal(u, v, equal_nan=True) for u, v in zip(r["tiled"], r["resident-log"])))
for B in (64, 65):
    rs = np.random.RandomState(100 + B)
    llr = _staggered(rs, B, 1944, 2.0 / 3, [2.0, 3.5, 30.0])
    llr[rs.randint(llr.size, size=40)] = 0.0
    llr[rs.randint(llr.size, size=10)] = -0.0
    llr[rs.randint(llr.size, size=10)] = 1e4
    llr[rs.randint(llr.size, size=10)] = -np.inf
    llr[1944 * 2 + 7] = np.nan
    r = {}
    for path in ("tiled", "resident"):
        d, o, i, x, k = _decode(_lib, path, llr, p, "SPA", 5)
        r[path] = (d, o, i)
    report("special B=%d" % B, r["tiled"], r["resident"])
_lib.ldpc_set_path(None)
