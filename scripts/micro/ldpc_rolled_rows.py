#!/usr/bin/env python3
"""Rolled check rows (13 .. 32 edges) on the tiled LDPC path: a random code with n_v = 4000, 500 checks of 24 .. 28 edges (13 000 edges),
B = 8192, pure-noise LLRs, 10 iterations, min-sum and sum-product.     CPX_LIB_PATH=... python scripts/micro/ldpc_rolled_rows.py <label>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from commpy_amd import _lib  # noqa: E402
from benchmarks.other_configs import Dev, time_steps  # noqa: E402


def main():
    from test_random_codes_gpu import _random_ldpc
    from commpy_amd.channelcoding.ldpc import _device_code
    label = sys.argv[1] if len(sys.argv) > 1 else "?"
    lib = _lib.load()
    rs = np.random.RandomState(77)
    n_v, n_c, B = 4000, 500, 8192
    p = _random_ldpc(rs, n_v, n_c, rs.randint(24, 29, size=n_c))
    code = _device_code(p)
    dev = Dev(lib)
    d_llr = dev.put(np.ascontiguousarray(rs.randn(B, n_v) * 2.0))
    d_dec, d_out, d_it = dev.empty(B * n_v), dev.empty(B * n_v * 8), dev.empty(B * 4)
    out = []
    _lib.ldpc_set_path("tiled")
    for alg, an in ((1, "MSA"), (0, "SPA")):
        ms = time_steps(lib, lambda: _lib.check(lib.cpx_ldpc_bp_decode_batch_bm_dev(code, d_llr, B, alg, 10, d_dec, d_out, d_it, None)), 5, 2)
        its = dev.get(d_it, (B,), np.int32)
        out.append("%s %.3f ms (it %.1f)" % (an, float(np.mean(ms)), its.mean()))
    print(label, "|", " | ".join(out), "|", _lib.last_kernel()[:60])
    dev.free()


if __name__ == "__main__":
    main()
