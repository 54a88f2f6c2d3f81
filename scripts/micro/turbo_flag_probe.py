#!/usr/bin/env python3
"""Debug aid (round 6): how many codewords of the config-3 workload raise a "detect and redo" flag, per iteration count and
codewords-per-pair geometry (cpx_last_kernel's redo suffix).  CPX_LIB_PATH selects the library build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np
from commpy_amd import _lib
from commpy_amd.channelcoding import turbo_decode
from benchmarks.other_configs import turbo_workload

for B in (16384, 4096):
    tr, il, msgs, s, p1, p2, nv = turbo_workload(B, 1024, 4)
    for it in (1, 2, 3, 6):
        dec = turbo_decode(s, p1, p2, tr, nv, it, il)
        k = _lib.last_kernel()
        print("B=%d its=%d BER %.3e  %s" % (B, it, (dec != msgs).mean(), k[k.find("redo"):][:60]))
