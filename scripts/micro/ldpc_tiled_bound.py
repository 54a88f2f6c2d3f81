#!/usr/bin/env python3
"""What bounds the tiled (beyond-LDS) sum-product path -- would the ratio-domain row pay there?  (VERDICT r04 missing 6.)
Config-4 code (1944,1296), sum-product, 1 dB (no block converges: every launch works on all B blocks), 10 iterations, on the three
path modes.  Run under `rocprofv3 --kernel-trace --stats` the per-kernel averages of the tiled pass kernels give their HBM rate:
    check pass   reads R (k > 0) and writes R: 2 E x 8 B per block, + Q once (n x 8 B; gathered from L2)
    variable pass reads R (E x 8 B) and the channel LLRs, writes Q (2 n x 8 B)
    python scripts/micro/ldpc_tiled_bound.py [B [paths [spa|msa]]]
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.other_configs import DESIGN_1944, Dev, time_steps     # noqa: E402
from commpy_amd import _lib                                                  # noqa: E402


def main():
    from commpy_amd.channelcoding.ldpc import _device_code, get_ldpc_code_params
    lib = _lib.load()
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    p = get_ldpc_code_params(DESIGN_1944, True)
    n, E = 1944, int(p["vnode_deg_list"].sum())
    rs = np.random.RandomState(3)
    sigma = 1.0
    llr0 = (2.0 / sigma ** 2) * (1.0 + sigma * rs.standard_normal((B, n)))       # all-zero codeword, BPSK, ~1 dB on the coded bit
    dev = Dev(lib)
    d_src = dev.put(np.ascontiguousarray(llr0))
    d_dec, d_out, d_it = dev.empty(B * n), dev.empty(B * n * 8), dev.empty(B * 4)
    code = _device_code(p)
    iters = 10
    alg = 1 if len(sys.argv) > 3 and sys.argv[3].lower() == "msa" else 0
    for path in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("tiled", "resident-log", "resident")):
        _lib.ldpc_set_path(path)
        try:
            def step():
                _lib.check(lib.cpx_ldpc_bp_decode_batch_bm_dev(code, d_src, B, alg, iters, d_dec, d_out, d_it, None))
            ms = time_steps(lib, step, 5, 2)
            its = dev.get(d_it, (B,), np.int32)
            rec = {"path": path, "alg": "MSA" if alg else "SPA", "kernel": _lib.last_kernel(), "B": B, "edges": E, "n": n, "iterations": iters, "ms": float(np.mean(ms)),
                   "mean_executed_iterations": float(its.mean()),
                   "tiled_bytes_per_iteration_GB": B * (3 * E + 3 * n) * 8 / 1e9}
            print(json.dumps(rec))
        finally:
            _lib.ldpc_set_path(None)
    dev.free()


if __name__ == "__main__":
    main()
