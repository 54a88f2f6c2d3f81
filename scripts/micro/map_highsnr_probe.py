#!/usr/bin/env python3
"""What the detect-and-redo of map_decode costs towards high SNR: B = 16384, N = 1024, 4-state RSC, BPSK +-1 + noise.
Measured (round 4, per-codeword flags): sigma^2 = 0.5 / 0.05: 0.34 / 0.35 ms, nothing flagged; sigma^2 = 0.02 and below: 28 ms --
EVERY codeword goes to the exact kernel, and rightly so: a wrong path of this code differs in >= 5 coded bits of e^-100 each, the
a-posteriori ratios app1 / app0 reach e^-700, i.e. the reference's own sums underflow and it returns -inf / imprecise LLRs there,
which only the literal absolute-scale kernel reproduces.  (Flag (A) itself fires for all pairs from sigma^2 = 0.0116 on.)"""
import ctypes
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from commpy_amd import _lib  # noqa: E402
from bench_kernels import Dev, timeit  # noqa: E402
from helpers import make_trellis  # noqa: E402

lib = _lib.load()
tr = make_trellis("rsc_legacy_4")
B, N = 16384, 1024
rs = np.random.RandomState(3)
dev = Dev(lib)
for nv in (0.5, 0.05, 0.02, 0.01):
    sy = rs.choice([-1.0, 1.0], size=(B, N)) + np.sqrt(nv) * rs.standard_normal((B, N))
    pa = rs.choice([-1.0, 1.0], size=(B, N)) + np.sqrt(nv) * rs.standard_normal((B, N))
    far = np.mean(np.any((np.abs(sy) + 1) ** 2 + (np.abs(pa) + 1) ** 2 > 345 * 2 * nv, axis=1))
    d_s, d_p, d_l = dev.put(sy), dev.put(pa), dev.put(np.zeros((B, N)))
    d_o, d_b = dev.empty(B * N * 8), dev.empty(B * N)
    h = tr._device_handle()
    ms, _ = timeit(lib, lambda: _lib.check(lib.cpx_map_decode_batch_dev(h, d_s, d_p, d_l, B, N, float(nv), 1, d_o, d_b, None)), steps=3)
    print("sigma^2 = %-5g codewords with a flag-(A) pair: %5.1f %%   map_decode %.2f ms" % (nv, 100 * far, ms), flush=True)
