#!/usr/bin/env python3
"""What the detect-and-redo of map_decode costs towards high SNR: B = 16384 valid codewords, N = 1024, 4-state RSC, BPSK + noise
(round 4, per-codeword flags).  Flag (A) -- a received pair whose worst branch probability is below e^-345 -- fires for every pair from
sigma^2 = 8 / 690 = 0.0116 on; above that only outliers raise it.  (Random +-1 symbols that are NOT a codeword are another matter: at
sigma^2 = 0.02 every parity contradiction shrinks the state metrics by e^-100 and flag (C) sends everything to the exact kernel: 28 ms.)"""
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
from commpy_amd.channelcoding import conv_encode_batch  # noqa: E402
coded = conv_encode_batch(rs.randint(0, 2, (B, N)), tr, "cont")
for nv in ([float(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else (0.5, 0.05, 0.02, 0.013, 0.01)):
    sy = 2.0 * coded[:, 0::2] - 1 + np.sqrt(nv) * rs.standard_normal((B, N))
    pa = 2.0 * coded[:, 1::2] - 1 + np.sqrt(nv) * rs.standard_normal((B, N))
    far = np.mean(np.any((np.abs(sy) + 1) ** 2 + (np.abs(pa) + 1) ** 2 > 345 * 2 * nv, axis=1))
    d_s, d_p, d_l = dev.put(sy), dev.put(pa), dev.put(np.zeros((B, N)))
    d_o, d_b = dev.empty(B * N * 8), dev.empty(B * N)
    h = tr._device_handle()
    ms, _ = timeit(lib, lambda: _lib.check(lib.cpx_map_decode_batch_dev(h, d_s, d_p, d_l, B, N, float(nv), 1, d_o, d_b, None)), steps=3)
    print("sigma^2 = %-5g codewords with a flag-(A) pair: %5.1f %%   map_decode %.2f ms" % (nv, 100 * far, ms), flush=True)
