#!/usr/bin/env python3
"""Sum-product rows against each other, by |LLR| band (the banded contract of helpers.spa_contract was loosened in
the same change that added the ratio-domain kernel -- these are the measurements the TIGHT row-vs-row bounds of round 5 are set from).

    python scripts/spa_rows_table.py [--out DIR]

  ratio vs log   ldpc_resident_ratio_kernel against the log-domain row (tiled path), the batches of tests/test_ldpc_resident_gpu.py
  log vs oracle  the log-domain rows ('resident-log', 'tiled') against the C oracle on the 128-block config-4 chain at 8 dB
Per band: values, max |dev|, fraction within 1e-5 / 1e-6 / 1e-8; separately for converged blocks (iterations < limit) and all blocks.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

BANDS = ((0.0, 10.0), (10.0, 26.0), (26.0, 50.0), (50.0, 1e9))


def table(a, b, conv_cols):
    """a, b: [n_v][B] LLRs; conv_cols: bool [B].  Rows of (band, which blocks, count, max dev, frac<=1e-5, 1e-6, 1e-8)."""
    rows = []
    for name, cols in (("converged", conv_cols), ("all", np.ones_like(conv_cols))):
        x, y = a[:, cols].ravel(), b[:, cols].ravel()
        fin = np.isfinite(x) & np.isfinite(y)
        dev, mag = np.abs(x[fin] - y[fin]), np.abs(y[fin])
        for lo, hi in BANDS:
            m = (mag >= lo) & (mag < hi)
            if m.any():
                d = dev[m]
                rows.append({"band": [lo, hi], "blocks": name, "values": int(m.sum()), "max_dev": float(d.max()),
                             "within_1e-5": float(np.mean(d <= 1e-5)), "within_1e-6": float(np.mean(d <= 1e-6)),
                             "within_1e-8": float(np.mean(d <= 1e-8)), "beyond_1e-5": int(np.sum(d > 1e-5)), "beyond_1e-6": int(np.sum(d > 1e-6))})
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import oracle
    from commpy_amd import _lib
    from helpers import ldpc_params
    from test_ldpc_resident_gpu import _decode, _staggered
    res = {}
    try:
        p = ldpc_params("n1944")
        rs = np.random.RandomState(5)
        llr = _staggered(rs, 2100, 1944, 2.0 / 3, [0.5, 2.0, 2.6, 3.2, 4.5, 30.0])
        for iters in (14, 50):
            d1, o1, i1, _, _ = _decode(_lib, "tiled", llr, p, "SPA", iters)
            d2, o2, i2, _, k2 = _decode(_lib, "resident", llr, p, "SPA", iters)
            res["ratio_vs_log n1944 B=2100 iters<=%d" % iters] = {"kernel": k2, "iterations_equal": bool(np.array_equal(i1, i2)),
                                                                  "dec_equal": bool(np.array_equal(d1, d2)), "rows": table(o2, o1, i1 < iters)}
        for name, n in (("gallager96", 96), ("wimax1440", 1440)):
            pp = ldpc_params(name)
            rs = np.random.RandomState(11)
            llr = _staggered(rs, 777, n, 0.5, [1.0, 2.5, 4.0, 30.0])
            d1, o1, i1, _, _ = _decode(_lib, "tiled", llr, pp, "SPA", 8)
            d2, o2, i2, _, _ = _decode(_lib, "resident", llr, pp, "SPA", 8)
            res["ratio_vs_log %s B=777 iters<=8" % name] = {"iterations_equal": bool(np.array_equal(i1, i2)), "rows": table(o2, o1, i1 < 8)}
            do, oo, io = oracle.ldpc_bp_decode(llr[:64 * n].copy(), pp, "SPA", 8, True)
            res["log_vs_oracle %s 64 blocks iters<=8" % name] = {"iterations_equal": bool(np.array_equal(i1[:64], io)), "rows": table(o1[:, :64], oo, io < 8)}
        # the 128-block config-4 chain at 8 dB (tests/test_config_sizes_gpu.py::test_config4_chain_128_blocks_vs_oracle)
        from commpy_amd.devicelink import LdpcEncoder
        from commpy_amd.modulation import QAMModem
        md = QAMModem(64)
        rs = np.random.RandomState(44)
        code = LdpcEncoder(p, "gf2").encode(rs.randint(0, 2, (128, 1296)))
        N0 = md.Es / ((2.0 / 3) * 6 * 10 ** 0.8)
        s = md.modulate(code.reshape(-1))
        y = s + np.sqrt(N0 / 2) * (rs.standard_normal(len(s)) + 1j * rs.standard_normal(len(s)))
        llr = -md.demodulate(y, "soft", N0)
        do, oo, io = oracle.ldpc_bp_decode(llr.copy(), p, "SPA", 50, True)
        for path in ("tiled", "resident-log", "resident"):
            d, o, i, _, k = _decode(_lib, path, llr, p, "SPA", 50)
            res["%s_vs_oracle config-4 chain 128 blocks 8 dB" % path] = {"kernel": k, "iterations_equal": bool(np.array_equal(i, io)),
                                                                          "dec_equal": bool(np.array_equal(d, do)), "rows": table(o, oo, io < 50)}
    finally:
        _lib.ldpc_set_path(None)
    for k, v in res.items():
        print(k, {kk: vv for kk, vv in v.items() if kk != "rows"})
        for r in v["rows"]:
            print("   ", r)
    if a.out:
        with open(os.path.join(a.out, "spa_rows_table.json"), "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
