#!/usr/bin/env python3
"""Round 6 probe: what the gathered rows of the time-major turbo slab cost -- config-3 decode time with the benchmark's random
interleaver against the identity permutation (rows then arrive in order), 4 and 8 states.  CPX_LIB_PATH selects the build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from commpy_amd import _lib
from benchmarks.bench_kernels import Dev, timeit
from benchmarks.other_configs import turbo_workload
lib = _lib.load()
for states in (4, 8):
    N, B = 1024, 16384
    tr, il, msgs, s, p1, p2, nv = turbo_workload(B, N, states)
    dev = Dev(lib)
    d_s, d_p1, d_p2 = dev.put(s), dev.put(p1), dev.put(p2)
    d_bits = dev.empty(B * N)
    h = tr._device_handle()
    for name, perm in (("random", np.asarray(il.p_array, dtype=np.int32)), ("identity", np.arange(N, dtype=np.int32)),
                       ("block-of-8 shuffle", (np.random.RandomState(1).permutation(N // 8)[:, None] * 8 + np.arange(8)).reshape(-1).astype(np.int32))):
        d_perm = dev.put(perm)
        ms, mn = timeit(lib, lambda: _lib.check(lib.cpx_turbo_decode_batch_dev(h, d_s, d_p1, d_p2, None, d_perm, B, N, nv, 6, d_bits, None)), steps=5)
        print("%d states, %-18s interleaver: %.3f ms (min %.3f)" % (states, name, ms, mn), flush=True)
    dev.free()
