import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, oracle
from helpers import *
from commpy_amd.channelcoding import ldpc_bp_decode
g = golden('ldpc')
p = ldpc_params('wimax1440')
fails = 0; n = 0
for rep in range(15):
    for alg in ('MSA', 'SPA'):
        for it in (1, 3, 5, 8):
            llr = g['l019__llr'].copy()
            d, o, its = ldpc_bp_decode(llr.copy(), p, alg, it, return_iterations=True)
            do, oo, io = oracle.ldpc_bp_decode(llr.copy(), p, alg, it, True)
            n += 1
            if not (np.array_equal(d, do) and np.array_equal(its, io) and np.nanmax(np.abs(o - oo)) < 1e-9):
                fails += 1
                print('FAIL rep', rep, alg, it, its, io, int(np.sum(d != do)))
print(os.environ.get('CPX_SYNC_ALLOC'), 'fails', fails, 'of', n)
