#!/usr/bin/env python3
"""Debug aid (round 6): fp32-fast vs fp64 turbo decode per codeword slot, for two codewords-per-pair geometries."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np
import commpy_amd
from commpy_amd import _lib
from commpy_amd.channelcoding import RandInterlv, turbo_decode
from commpy_amd.devicelink import turbo_encode_gpu
from helpers import make_trellis

tr = make_trellis("rsc_legacy_4")
for B, N, iters in ((4096, 1024, 6), (16384, 256, 2), (1024, 64, 1)):
    rs = np.random.RandomState(1)
    il = RandInterlv(N, 99)
    msgs = rs.randint(0, 2, (B, N))
    nv = 1 / (2 * (1.0 / 3) * 10 ** (1.5 / 10.0))
    s, p1, p2 = (a[:, :N] * 2.0 - 1 + np.sqrt(nv) * rs.standard_normal((B, N)) for a in turbo_encode_gpu(msgs, tr, tr, il))
    ref = turbo_decode(s, p1, p2, tr, nv, iters, il)
    print(_lib.last_kernel())
    commpy_amd.set_precision("fp32-fast")
    fast = turbo_decode(s, p1, p2, tr, nv, iters, il)
    print(_lib.last_kernel())
    commpy_amd.set_precision("fp64-parity")
    bad = (fast != ref).mean(1)
    gw = int(_lib.last_kernel().split("wave pairs per workgroup, ")[1].split(" codewords")[0]) if "codewords per pair" in _lib.last_kernel() else 16
    print("B=%d N=%d: BER ref %.3e fast %.3e; bad codewords %d; by slot:" % (B, N, (ref != msgs).mean(), (fast != msgs).mean(), (bad > 0.01).sum()),
          [int((bad[g::gw] > 0.01).sum()) for g in range(gw)])
    cols = (fast != ref).mean(0)
    print("   by time position (first 16):", np.round(cols[:16], 2), " mean over t%8:", [round(float(cols[j::8].mean()), 3) for j in range(8)])
