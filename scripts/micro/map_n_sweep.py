#!/usr/bin/env python3
"""Fixed cost of a MAP pass launch: map_decode at B = 16384 over block lengths N (time is a + b*N; `a` is what a launch costs
besides the recursions: dispatch, table loads, the checkpoint hand-over, the end-of-kernel write-back)."""
import os, sys, json, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "benchmarks"))
from commpy_amd import _lib
from bench_kernels import Dev, timeit
from commpy_amd.channelcoding import Trellis

lib = _lib.load()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    tr = Trellis(np.array([2]), np.array([[1, 7]]), 5, "rsc")
h = tr._device_handle()
B = 16384
rs = np.random.RandomState(3)
res = []
for N in (8, 16, 64, 256, 512, 1024, 2048):
    s = rs.randint(0, 2, (B, N)) * 2.0 - 1 + 0.8 * rs.randn(B, N)
    p = rs.randint(0, 2, (B, N)) * 2.0 - 1 + 0.8 * rs.randn(B, N)
    dev = Dev(lib)
    d_s, d_p, d_z = dev.put(s), dev.put(p), dev.put(np.zeros((B, N)))
    d_L, d_bits = dev.empty(B * N * 8), dev.empty(B * N)
    ms, _ = timeit(lib, lambda: _lib.check(lib.cpx_map_decode_batch_dev(h, d_s, d_p, d_z, B, N, 0.64, 1, d_L, d_bits, None)), steps=5)
    res.append((N, ms))
    print(json.dumps({"N": N, "ms": round(ms, 4), "us_per_chunk": round(ms * 1e3 / max(N // 8, 1), 3)}))
    dev.free()
