#!/usr/bin/env python3
"""K = 3 (5,7) and K = 5 (23,35): the small-ring fused kernel against the state-per-lane kernels at chip-filling batches."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
from commpy_amd import _lib
from bench_kernels import Dev, timeit
from commpy_amd.channelcoding import Trellis, conv_encode_batch
lib = _lib.load()
rs = np.random.RandomState(3)
for mem, gm, B, nbits in ((2, [5, 7], 1 << 20, 64), (2, [5, 7], 1 << 17, 1024), (4, [0o23, 0o35], 1 << 18, 256), (4, [0o23, 0o35], 1 << 16, 1024)):
    tr = Trellis(np.array([mem]), np.array([gm]))
    coded = conv_encode_batch(rs.randint(0, 2, (B, nbits)).astype(np.uint8), tr).astype(np.float64)
    rx = np.where(rs.random_sample(coded.shape) < 0.05, 1 - coded, coded)
    L = nbits + mem
    dev = Dev(lib)
    d_in, d_out = dev.put(rx), dev.empty(B * L)
    h = tr._device_handle()
    for path in (None, "wave"):
        _lib.viterbi_set_path(path)
        ms, _ = timeit(lib, lambda: _lib.check(lib.cpx_viterbi_decode_batch_dev(h, d_in, B, 2 * L, L, L, 5 * mem, 0, d_out, None)), steps=5)
        print(gm, B, nbits, path or "auto", round(ms, 3), "ms", _lib.last_kernel(), flush=True)
    _lib.viterbi_set_path(None)
    dev.free()

# table-driven small codes: K = 4 (15,17), K = 6 (53,75)
for mem, gm, B, nbits in ((3, [0o15, 0o17], 1 << 18, 256), (5, [0o53, 0o75], 1 << 16, 1024)):
    tr = Trellis(np.array([mem]), np.array([gm]))
    coded = conv_encode_batch(rs.randint(0, 2, (B, nbits)).astype(np.uint8), tr).astype(np.float64)
    rx = np.where(rs.random_sample(coded.shape) < 0.05, 1 - coded, coded)
    L = nbits + mem
    dev = Dev(lib)
    d_in, d_out = dev.put(rx), dev.empty(B * L)
    h = tr._device_handle()
    for path in (None, "wave"):
        _lib.viterbi_set_path(path)
        ms, _ = timeit(lib, lambda: _lib.check(lib.cpx_viterbi_decode_batch_dev(h, d_in, B, 2 * L, L, L, 5 * mem, 0, d_out, None)), steps=5)
        print(gm, B, nbits, path or "auto", round(ms, 3), "ms", _lib.last_kernel(), flush=True)
    _lib.viterbi_set_path(None)
    dev.free()
