#!/usr/bin/env python3
"""The table-driven fused Viterbi kernel (branch metrics selected by VGPR index mode, round 5) on codes of 8 .. 64 states: 65 536 x 1024-bit
blocks, soft, default traceback depth; and the state-per-lane kernels on the same input."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from commpy_amd import _lib  # noqa: E402
from benchmarks.other_configs import Dev, time_steps  # noqa: E402
from commpy_amd.channelcoding import Trellis, conv_encode_batch  # noqa: E402

lib = _lib.load()
B = 65536
rs = np.random.RandomState(4)
for mem, gm in ((6, [0o135, 0o147]), (5, [0o53, 0o75]), (4, [0o23, 0o35]), (3, [0o15, 0o17])):
    tr = Trellis(np.array([mem]), np.array([gm]))
    coded = conv_encode_batch(rs.randint(0, 2, (B, 1024)).astype(np.uint8), tr).astype(np.float64)
    llr = np.ascontiguousarray(4.0 * coded - 2 + rs.standard_normal(coded.shape).astype(np.float32) * 1.4, dtype=np.float64)
    L = coded.shape[1] // 2
    dev = Dev(lib)
    d_in, d_out = dev.put(llr), dev.empty(B * L)
    h = tr._device_handle()
    out = []
    for path in (None, "wave"):
        _lib.viterbi_set_path(path)
        try:
            ms = time_steps(lib, lambda: _lib.check(lib.cpx_viterbi_decode_batch_dev(h, d_in, B, coded.shape[1], L, L + mem - 1, min(5 * mem, L), 1, d_out, None)), 10, 3)
            out.append((float(np.median(ms)), _lib.last_kernel()))
        finally:
            _lib.viterbi_set_path(None)
    print("K=%d (%o,%o): %.3f ms [%s]   state-per-lane %.3f ms" % (mem + 1, gm[0], gm[1], out[0][0], out[0][1], out[1][0]), flush=True)
    dev.free()
