#!/usr/bin/env python3
"""Table-driven exp / log in the separable soft demodulator against the library functions ("libm" path mode), same process, alternated:
64-QAM at 8 dB (10.6 M symbols, the config-4 workload) and 256-QAM at 14 dB; ms per call, max |LLR difference| between the two modes over the
whole array, and both against the oracle on the first 20 000 symbols.     python scripts/micro/demod_tab_ab.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from commpy_amd import _lib  # noqa: E402
from benchmarks.other_configs import Dev, time_steps  # noqa: E402


def main():
    import oracle
    from commpy_amd.modulation import QAMModem
    lib = _lib.load()
    for m, ns, snr_db in ((64, 32768 * 324, 8.0), (256, 8_000_000, 14.0)):
        md = QAMModem(m)
        rs = np.random.RandomState(31)
        N0 = md.Es / ((2.0 / 3) * md.num_bits_symbol * 10 ** (snr_db / 10.0))
        y = md.constellation[rs.randint(0, m, ns)] + np.sqrt(N0 / 2) * (rs.randn(ns) + 1j * rs.randn(ns))
        dev = Dev(lib)
        d_y = dev.put(y)
        nb = md.num_bits_symbol
        d_l = dev.empty(ns * nb * 8)
        h = md._device_handle()
        want = oracle.demodulate(md.constellation, y[:20000], "soft", N0)
        res, outs = {}, {}
        for rnd in range(2):
            for mode in ("libm", None):
                _lib.demod_set_path(mode)
                try:
                    ms = time_steps(lib, lambda: _lib.check(lib.cpx_demod_soft_dev(h, d_y, ns, float(N0), d_l, None)), 10, 3)
                    kern = _lib.last_kernel()
                    out = dev.get(d_l, (ns * nb,), np.float64)
                finally:
                    _lib.demod_set_path(None)
                key = mode or "auto"
                res.setdefault(key, []).append(round(float(np.mean(ms)), 4))
                outs[key] = (out, kern)
        a, b = outs["auto"][0], outs["libm"][0]
        fin = np.isfinite(a) & np.isfinite(b)
        print(json.dumps({"M": m, "symbols": ns, "snr_db": snr_db, "ms": res, "kernels": {k: v[1] for k, v in outs.items()},
                          "max_abs_diff_between_modes": float(np.max(np.abs(a[fin] - b[fin]))), "same_nonfinite_pattern": bool(np.array_equal(np.isfinite(a), np.isfinite(b))),
                          "max_abs_err_vs_oracle": {k: float(np.max(np.abs(v[0][:20000 * nb] - want))) for k, v in outs.items()},
                          "GBps_auto": ns * (16 + 8 * nb) / (min(res["auto"]) * 1e-3) / 1e9}))
        dev.free()


if __name__ == "__main__":
    main()
