#!/usr/bin/env python3
"""LDS bank-conflict model of the LDS-resident LDPC kernel (csrc/ldpc_resident.hip) for one block-iteration of a given code:
extra LDS-array cycles of the Q gathers of the check pass and the R gathers of the variable pass, for the kernel's layouts.
Lane groups / banks as in scripts/micro/lds_bank_sim.py (MI355X_MICROARCH.md, LDS).  Host arithmetic only.

    python scripts/micro/ldpc_bank_sim.py [design file]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts", "micro"))
from lds_bank_sim import G_R64, extra_cycles  # noqa: E402


def tables(path):
    from commpy_amd.channelcoding.ldpc import get_ldpc_code_params, _edge_list
    p = get_ldpc_code_params(path, True)
    ec, ev = _edge_list(p)
    n_v, n_c = int(p["n_vnodes"]), int(p["n_cnodes"])
    rows = [[] for _ in range(n_c)]
    for c, v in zip(ec, ev):
        rows[c].append(int(v))
    cols = [[] for _ in range(n_v)]
    for c in range(n_c):
        for j, v in enumerate(rows[c]):
            cols[v].append((c, j))
    return n_v, n_c, rows, cols


def simulate(n_v, n_c, rows, cols, q_addr, r_addr, check_threads, var_threads, what):
    """q_addr(c, j) -> byte address of the Q gathered by edge j of check c; r_addr(v, q) -> address of the q-th R of variable v;
    check_threads / var_threads: lists of waves, each a list of 64 node indices (or -1 = idle lane)."""
    cdeg = max(len(r) for r in rows)
    vdeg = max(len(c) for c in cols)
    tot_i = tot_e = 0
    for wave in check_threads:
        for j in range(cdeg):
            addrs = [q_addr(c, j) if c >= 0 and j < len(rows[c]) else None for c in wave]
            if all(a is None for a in addrs):
                continue
            tot_i += 1
            tot_e += extra_cycles(addrs, G_R64, 8, 64)
    ci, ce = tot_i, tot_e
    for wave in var_threads:
        for q in range(((vdeg + 3) // 4) * 4):
            addrs = [r_addr(v, q) if v >= 0 and q < len(cols[v]) else None for v in wave]
            if all(a is None for a in addrs):
                continue
            tot_i += 1
            tot_e += extra_cycles(addrs, G_R64, 8, 64)
    print("%-46s check pass: %4d gathers, %4d extra cycles (%.2f per instr); variable pass: %4d gathers, %4d extra (%.2f)"
          % (what, ci, ce, ce / max(ci, 1), tot_i - ci, tot_e - ce, (tot_e - ce) / max(tot_i - ci, 1)))
    return tot_i, tot_e


def waves_of(order):
    order = list(order)
    while len(order) % 64:
        order.append(-1)
    return [order[i:i + 64] for i in range(0, len(order), 64)]


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "commpy_amd/channelcoding/designs/ldpc/ieee80211n/1944.1296.txt")
    n_v, n_c, rows, cols = tables(path)
    rs = max(len(r) for r in rows) | 1
    roff = ((n_v + 2) & ~1) * 8
    print("code: n_v %d n_c %d, row stride %d" % (n_v, n_c, rs))
    # round 3 / 4 layout: Q[v] at 8 v, R[c][j] at roff + 8 (c rs + j), thread = node
    simulate(n_v, n_c, rows, cols, lambda c, j: 8 * rows[c][j], lambda v, q: roff + 8 * (cols[v][q][0] * rs + cols[v][q][1]),
             waves_of(range(n_c)), waves_of(range(n_v)), "current layout, thread = node")
    # QC structure: Z = 81; pad every block row / column to 96 lanes (no 32-lane group straddles two blocks)
    for Z in (81,):
        if n_v % Z or n_c % Z:
            continue
        Zp = (Z + 31) // 32 * 32

        def padded(n):
            out = []
            for b in range(n // Z):
                out += list(range(b * Z, b * Z + Z)) + [-1] * (Zp - Z)
            return out
        simulate(n_v, n_c, rows, cols, lambda c, j: 8 * rows[c][j], lambda v, q: roff + 8 * (cols[v][q][0] * rs + cols[v][q][1]),
                 waves_of(padded(n_c)), waves_of(padded(n_v)), "blocks padded to %d lanes" % Zp)
        # + mirrored Q: block column j at Zq j, entries Z .. Z + 30 repeat 0 .. 30; an edge reads the copy that keeps its 32-lane
        # group contiguous
        Zq = Z + 31 + ((Z + 31) & 1)

        def q_addr_m(c, j):
            v = rows[c][j]
            bj, x = divmod(v, Z)
            r = c % Z
            g0 = (r // 32) * 32                                   # first lane of this lane group inside the block row
            # the group's first check reads x0; this lane's value continues that run if x < x0 (wrapped)
            v0 = rows[(c // Z) * Z + g0][j] if j < len(rows[(c // Z) * Z + g0]) else None
            x0 = v0 % Z if v0 is not None and v0 // Z == bj else None
            if x0 is not None and x < x0 and x < 31:
                x += Z                                            # mirror copy
            return 8 * (bj * Zq + x)
        simulate(n_v, n_c, rows, cols, q_addr_m, lambda v, q: roff + 8 * (cols[v][q][0] * rs + cols[v][q][1]),
                 waves_of(padded(n_c)), waves_of(padded(n_v)), "padded + mirrored Q (31 entries per column)")
        # + R position-major with mirrored ... (costed only): R[pos][c] with the same mirror trick per block row
        Zr = Zq
        nbr = n_c // Z

        def r_addr_m(v, q):
            c, pos = cols[v][q]
            bi, y = divmod(c, Z)
            r = v % Z
            g0 = (r // 32) * 32
            v0 = (v // Z) * Z + g0
            c0 = cols[v0][q][0] if q < len(cols[v0]) else None
            y0 = c0 % Z if c0 is not None and c0 // Z == bi else None
            if y0 is not None and y < y0 and y < 31:
                y += Z
            return roff + 8 * ((pos * nbr + bi) * Zr + y)
        simulate(n_v, n_c, rows, cols, q_addr_m, r_addr_m, waves_of(padded(n_c)), waves_of(padded(n_v)),
                 "padded + mirrored Q + position-major mirrored R")
