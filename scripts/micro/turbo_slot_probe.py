#!/usr/bin/env python3
"""Debug aid (round 6): turbo_decode against the CPU oracle per codeword slot for full pairs (B = 16384) at several block lengths."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np
import oracle
from commpy_amd.channelcoding import turbo_decode
from helpers import make_trellis, Perm
tr = make_trellis("rsc_legacy_4")
for N, its in ((64, 1), (256, 1), (64, 3)):
    rs = np.random.RandomState(N)
    B = 16384
    perm = rs.permutation(N)
    s, p1, p2 = (np.sign(rs.randn(B, N)) + 0.9 * rs.randn(B, N) for _ in range(3))
    dec = turbo_decode(s, p1, p2, tr, 0.81, its, Perm(perm))
    bad = np.zeros(16, int); pos = np.zeros(N, int); n = 0
    for b in range(0, 2048):
        want = oracle.turbo_decode(s[b], p1[b], p2[b], tr, 0.81, its, Perm(perm))
        d = dec[b] != want
        if d.any():
            bad[b % 16] += 1; pos += d; n += 1
    print("N=%d its=%d: %d of 2048 codewords differ from the oracle; by slot %s; by position %s" % (N, its, n, bad.tolist(), np.nonzero(pos)[0][:24].tolist()))
