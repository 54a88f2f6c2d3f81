#!/usr/bin/env python3
"""Round 6, review item 9: search, in the LDS bank model of scripts/micro/ldpc_bank_sim.py, for layouts of the LDS-resident LDPC kernel
that cost ZERO extra instructions -- per-block-column offsets of the Q slots (a thread's three Q addresses can live in registers for the
whole launch) and per-block-row offsets of the R rows (folded into the column tables of ready-made addresses), found by coordinate
descent on the model's extra LDS cycles per block-iteration of the 802.11n (1944,1296) code.  Host arithmetic only.
Result (profiles/r06_ldpc_bank_search.txt): check-pass gathers 102 -> 90 extra cycles, variable-pass gathers 116 -> 114."""
import os, sys, random
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'scripts','micro'))
from lds_bank_sim import G_R64, extra_cycles
from ldpc_bank_sim import tables, waves_of
path=os.path.join(ROOT,"commpy_amd/channelcoding/designs/ldpc/ieee80211n/1944.1296.txt")
n_v,n_c,rows,cols=tables(path)
Z=81
rs=max(len(r) for r in rows)|1
cdeg=max(len(r) for r in rows); vdeg=max(len(c) for c in cols)
print('rs',rs,'cdeg',cdeg,'vdeg',vdeg, 'col degs', sorted(set(len(c) for c in cols)))
NBC=n_v//Z; NBR=n_c//Z
cw=waves_of(range(n_c)); vw=waves_of(range(n_v))
def check_cost(dq):
    tot=0
    for wave in cw:
        for j in range(cdeg):
            addrs=[(8*(rows[c][j]+dq[rows[c][j]//Z]) if c>=0 and j<len(rows[c]) else None) for c in wave]
            if all(a is None for a in addrs): continue
            tot+=extra_cycles(addrs,G_R64,8,64)
    return tot
def var_cost(dr):
    tot=0
    for wave in vw:
        for q in range(((vdeg+3)//4)*4):
            addrs=[(8*((cols[v][q][0]*rs+cols[v][q][1])+dr[cols[v][q][0]//Z]) if v>=0 and q<len(cols[v]) else None) for v in wave]
            if all(a is None for a in addrs): continue
            tot+=extra_cycles(addrs,G_R64,8,64)
    return tot
dq=[0]*NBC; dr=[0]*NBR
print('base check',check_cost(dq),'var',var_cost(dr))
random.seed(1)
best=check_cost(dq)
for it in range(3):
    for b in range(NBC):
        bb=dq[b]; bc=best
        for d in range(32):
            dq[b]=d; c=check_cost(dq)
            if c<bc: bc=c; bb=d
        dq[b]=bb; best=bc
    print('iter',it,'check',best,dq)
bestv=var_cost(dr)
for it in range(3):
    for b in range(NBR):
        bb=dr[b]; bc=bestv
        for d in range(32):
            dr[b]=d; c=var_cost(dr)
            if c<bc: bc=c; bb=d
        dr[b]=bb; bestv=bc
    print('iter',it,'var',bestv,dr)
