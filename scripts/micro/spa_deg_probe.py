#!/usr/bin/env python3
"""Sum-product on random high-degree Tanner graphs against the oracle: which criterion fails, by how much, on what message size."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from test_random_codes_gpu import _random_ldpc
from commpy_amd import _lib
from commpy_amd.channelcoding import ldpc_bp_decode
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
nfail = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 600):
    n_c = int(rs.randint(8, 120)); n_v = int(n_c + rs.randint(8, 200))
    hi = int(min(31, n_v - 1, rs.randint(14, 32))); lo = int(rs.randint(max(2, hi - 4), hi + 1))
    try:
        p = _random_ldpc(rs, n_v, n_c, rs.randint(lo, hi + 1, size=n_c))
    except Exception:
        continue
    B = int(rs.choice([1, 3]))
    llr = rs.randn(B * n_v) * rs.choice([1.0, 3.0]) + rs.choice([0.0, 1.5, 4.0])
    llr[rs.randint(0, llr.size, 6)] = 0.0
    if np.max(np.abs(llr)) > 12.0:
        continue
    iters = int(rs.randint(1, 4))
    try:
        do, oo, io = oracle.ldpc_bp_decode(llr.copy(), dict(p), "SPA", iters, True)
        d, o, it = ldpc_bp_decode(llr.copy(), dict(p), "SPA", iters, return_iterations=True)
    except ValueError:
        continue
    fin = np.isfinite(oo)
    dev = np.abs(o[fin] - oo[fin]); mag = np.abs(oo[fin])
    bad = dev > 1e-5 + 1e-6 * mag
    nan_dec = int(np.sum(d[~fin] != do[~fin]))
    if not np.array_equal(it, io) or not np.array_equal(np.isfinite(o), fin) or bad.any() or not np.array_equal(d, do):
        nfail += 1
        print("case", case, "n_v", n_v, "n_c", n_c, "deg", lo, hi, "B", B, "iters", iters, "its equal", np.array_equal(it, io),
              "finite equal", np.array_equal(np.isfinite(o), fin), "dec equal", np.array_equal(d, do), "dec != at non-finite", nan_dec, "of", int(np.sum(~fin)), "n bad", int(bad.sum()),
              "max dev", float(dev.max()) if dev.size else 0, "at |LLR|", float(mag[np.argmax(dev)]) if dev.size else 0,
              "max |LLR|", float(mag.max()) if mag.size else 0, flush=True)
print("failures", nfail)
