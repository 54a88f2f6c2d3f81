#!/usr/bin/env python3
"""Config-2 geometry (65536 codewords, 1024-bit blocks, soft) with a K = 7 code that has no compiled-in instantiation: time of the
table-driven fused kernel against the state-per-lane kernels, and against (133,171) through the compiled-in kernel."""
import os, sys, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
from commpy_amd import _lib
from bench_kernels import Dev, timeit
from commpy_amd.channelcoding import Trellis, conv_encode_batch
lib = _lib.load()
rs = np.random.RandomState(3)
B = 65536
for gm in ([0o135, 0o147], [0o133, 0o171]):
    tr = Trellis(np.array([6]), np.array([gm]))
    coded = conv_encode_batch(rs.randint(0, 2, (B, 1024)).astype(np.uint8), tr).astype(np.float64)
    llr = 4.0 * coded - 2 + rs.standard_normal(coded.shape).astype(np.float32) * 1.4
    dev = Dev(lib)
    d_in, d_out = dev.put(np.ascontiguousarray(llr, dtype=np.float64)), dev.empty(B * 1030)
    h = tr._device_handle()
    for path in (None, "wave"):
        _lib.viterbi_set_path(path)
        ms, _ = timeit(lib, lambda: _lib.check(lib.cpx_viterbi_decode_batch_dev(h, d_in, B, 2060, 1030, 1030, 30, 1, d_out, None)), steps=5)
        print([oct(g) for g in gm], path or "auto", round(ms, 3), "ms", _lib.last_kernel(), flush=True)
    _lib.viterbi_set_path(None)
    dev.free()
