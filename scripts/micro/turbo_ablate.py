#!/usr/bin/env python3
"""Timing leg of the turbo-pass ablation (round 5): config 3 (B = 16384, N = 1024, 4 states, 6 iterations) on whichever build CPX_LIB_PATH
names -- the shipped library, or throw-away builds of csrc/bcjr.hip whose pass kernel (a) takes its three inputs from registers instead of
memory, (b) has its output stores masked off, (c) both.  Prints ms per decode and the kernel string (a "redo:" part says how many codewords
the garbage inputs sent to the literal kernel: that time is not the pass's).     python scripts/micro/turbo_ablate.py <label>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from commpy_amd import _lib  # noqa: E402
from benchmarks.other_configs import Dev, time_steps, turbo_workload  # noqa: E402


def main():
    lib = _lib.load()
    B, N = 16384, 1024
    tr, il, msgs, s, p1, p2, nv = turbo_workload(B, N)
    dev = Dev(lib)
    d_s, d_p1, d_p2 = dev.put(s), dev.put(p1), dev.put(p2)
    d_perm = dev.put(np.asarray(il.p_array, dtype=np.int32))
    d_bits = dev.empty(B * N)
    h = tr._device_handle()
    ms = time_steps(lib, lambda: _lib.check(lib.cpx_turbo_decode_batch_dev(h, d_s, d_p1, d_p2, None, d_perm, B, N, nv, 6, d_bits, None)), 6, 3)
    print(sys.argv[1] if len(sys.argv) > 1 else "?", "%.3f / %.3f ms" % (float(np.mean(ms)), float(np.min(ms))), "|", _lib.last_kernel()[:160])
    dev.free()


if __name__ == "__main__":
    main()
