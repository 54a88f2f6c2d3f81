#!/usr/bin/env python3
"""Ratio-domain sum-product kernel against the log-domain row AND the C oracle where belief propagation is chaotic: LLRs scaled up
(a receiver that over-estimates its SNR), blocks that never converge.  Prints, per regime, how many blocks differ in iteration count
or dec_word between (ratio, log row), (ratio, oracle), (log row, oracle).       python scripts/micro/spa_ratio_stress.py  (GPU box)"""
import sys

import numpy as np

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import oracle  # noqa: E402
from helpers import ldpc_params  # noqa: E402
from commpy_amd import _lib  # noqa: E402
from commpy_amd.channelcoding import ldpc_bp_decode  # noqa: E402

p = ldpc_params("n1944")
n = 1944
rs = np.random.RandomState(2)


def blocks_differ(a, b):
    (d1, o1, i1), (d2, o2, i2) = a, b
    with np.errstate(invalid="ignore"):
        ok = ~(np.isnan(o1) | np.isnan(o2))
        bad = (i1 != i2) | np.any((d1 != d2) & ok, axis=0) | np.any(np.isnan(o1) != np.isnan(o2), axis=0)
    return int(bad.sum())


for ebn0, scale, B in ((1.0, 1.0, 48), (1.0, 2.0, 48), (1.0, 4.0, 48), (2.2, 1.5, 48), (2.2, 3.0, 48), (3.0, 2.0, 48), (3.0, 4.0, 48), (3.0, 8.0, 48),
                       (2.2, 30.0, 24), (1.0, 100.0, 24)):
    sigma = 1 / np.sqrt(10 ** (ebn0 / 10.0) * (2.0 / 3) * 2)
    llr = (scale * 2.0 * (1.0 + sigma * rs.randn(B, n)) / sigma ** 2).reshape(-1)
    res = {}
    for path in ("resident", "tiled"):
        _lib.ldpc_set_path(path)
        d, o, i = ldpc_bp_decode(llr.copy(), p, "SPA", 50, return_iterations=True)
        res[path] = (d, o, i)
    _lib.ldpc_set_path(None)
    do, oo, io = oracle.ldpc_bp_decode(llr.copy(), p, "SPA", 50, True)
    res["oracle"] = (do, oo, io)
    print("Eb/N0 %.1f dB, LLRs x %-5g  %2d blocks, converged (oracle) %2d: differ ratio/log %2d, ratio/oracle %2d, log/oracle %2d" % (
        ebn0, scale, B, int((io < 50).sum()), blocks_differ(res["resident"], res["tiled"]), blocks_differ(res["resident"], res["oracle"]),
        blocks_differ(res["tiled"], res["oracle"])))
