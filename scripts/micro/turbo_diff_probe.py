#!/usr/bin/env python3
"""Debug aid (round 6): decode the config-3 workload with n iterations and save the bits (CPX_LIB_PATH selects the library);
with --diff a b prints where two saved decodes differ (codeword slot within its pair of wavefronts, time position)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np
if sys.argv[1] == "--diff":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    d = a != b
    cw = np.nonzero(d.any(1))[0]
    print("codewords differing:", len(cw), "slots (cw % 16):", np.bincount(cw % 16, minlength=16).tolist())
    print("pairs (cw // 16) first 20:", sorted(set((cw // 16).tolist()))[:20])
    pos = np.nonzero(d.any(0))[0]
    print("positions differing:", len(pos), pos[:40].tolist())
    sys.exit(0)
from commpy_amd.channelcoding import turbo_decode
from benchmarks.other_configs import turbo_workload
tr, il, msgs, s, p1, p2, nv = turbo_workload(16384, 1024, 4)
np.save(sys.argv[2], turbo_decode(s, p1, p2, tr, nv, int(sys.argv[1]), il).astype(np.uint8))
print("perm[:8]", il.p_array[:8])
