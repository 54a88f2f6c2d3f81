#!/usr/bin/env python3
"""Host model of the RATIO-DOMAIN sum-product row (round 4 experiment, then csrc/ldpc_resident.hip): the decoder state is kept as
likelihood ratios X_v = exp(out_llr_v), rho_cj = exp(R_cj), so an iteration has no exp and no log at all:

    x_j   = X[v_j] / rho_j                    (= exp(m_j), m_j the variable -> check message, ldpc.py:244-245)
    e_j   = min(X, rho) / max(X, rho)         (= exp(-|m_j|): ONE division), sign(m_j) = X >= rho
    rho_j <- (W u_j + U w_j) / (W u_j - U w_j) (= exp(2 atanh(prod_{i != j} tanh(m_i / 2))), the one-division row of ldpc_dev.h)
    X_v   <- exp(llr_v) * prod_j rho_j
rows near saturation take the exact-order sequence in the log domain exactly as before (and store exp of the result); out_llrs of a
retired block are llr + sum_j log(rho_j) over the rho its last VARIABLE pass used (the kernel returns log X, the same number, and lets a
saturated slot carry the sum itself).  This script decodes the live-reference blocks of tests/golden/ldpc_c4y.npz / ldpc_c4x.npz with
that arithmetic in NumPy float64 and checks dec_word, the iteration counts (against the C oracle) and the banded out_llrs contract
(tests/helpers.py spa_contract) -- i.e. whether the reformulation is numerically admissible -- before any kernel is written.

    python scripts/micro/spa_ratio_emul.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

XLO, XHI = 1e-290, 1e290


def decode(llr, ec, ev, n_c, n_v, n_iters, stats=None):
    """llr [B, n_v] -> dec [B, n_v], out [B, n_v], iters [B]; edges sorted by (check, variable)."""
    B = llr.shape[0]
    E = len(ec)
    row_start = np.searchsorted(ec, np.arange(n_c))
    order_v = np.lexsort((ec, ev))                                # edges grouped by variable, increasing check
    col_start = np.searchsorted(ev[order_v], np.arange(n_v))
    llr = np.clip(llr, -500.0, 500.0)
    with np.errstate(all="ignore"):
        E0 = np.exp(llr)
        X = np.clip(E0, XLO, XHI)
        rho = np.ones((B, E))
        active = np.ones(B, bool)
        iters = np.zeros(B, np.int32)
        for k in range(n_iters):
            a = np.nonzero(active)[0]
            if a.size == 0:
                break
            Xa, ra = X[a], rho[a]
            xe = Xa[:, ev]
            neg = (xe < 1.0)
            synd = np.bitwise_xor.reduceat(neg.astype(np.uint8), row_start, axis=1)
            done = ~synd.any(axis=1)
            active[a[done]] = False
            a = a[~done]
            if a.size == 0:
                break
            iters[a] += 1
            xe, ra = xe[~done], ra[~done]
            mn, mx = np.minimum(xe, ra), np.maximum(xe, ra)
            e = mn / mx
            sg = np.where(xe >= ra, 1.0, -1.0)
            u, w = sg * (1.0 - e), 1.0 + e
            U = np.multiply.reduceat(u, row_start, axis=1)
            W = np.multiply.reduceat(w, row_start, axis=1)
            emax = np.maximum.reduceat(e, row_start, axis=1)
            near = ~(np.abs(U) * (1.0 + emax) < np.abs(W) * (1.0 - emax) * (1.0 - 2.0 ** -32))
            Ue, We, ne = U[:, ec], W[:, ec], near[:, ec]
            n1, n2 = We * u, Ue * w
            new = (n1 + n2) / (n1 - n2)
            # near rows: the exact-order sequence (ldpc_dev.h spa_exact_t / spa_out_exact), result exponentiated
            t = sg * ((1.0 - e) / (1.0 + e))
            prod = np.multiply.reduceat(np.where(ne, t, 1.0), row_start, axis=1)[:, ec]
            x = np.clip((1.0 / t) * prod, -1.0, 1.0)
            R = np.clip(np.log((1.0 + x) / (1.0 - x)), -500.0, 500.0)
            new = np.where(ne, np.exp(R), new)
            if stats is not None:
                stats["rows"] += near.size
                stats["near"] += int(near.sum())
                stats["blocks_it"] += near.shape[0]
                stats["blocks_near"] += int(near.any(axis=1).sum())
                if "maxnear" in stats:
                    stats["maxnear"][a] = np.maximum(stats["maxnear"][a], near.sum(axis=1))
            rho[a] = new
            # variable pass: exp(llr) * prod rho, range-safe (big / small factors apart; else through logarithms)
            rv = new[:, order_v]
            big = np.multiply.reduceat(np.maximum(rv, 1.0), col_start, axis=1) * np.maximum(E0[a], 1.0)
            small = np.multiply.reduceat(np.minimum(rv, 1.0), col_start, axis=1) * np.minimum(E0[a], 1.0)
            Xn = np.clip(big * small, XLO, XHI)
            slow = ~np.isfinite(big) | (small < XLO)
            if slow.any():
                Q = llr[a] + np.add.reduceat(np.log(rv), col_start, axis=1)
                Xn = np.where(slow, np.clip(np.exp(Q), XLO, XHI), Xn)
                if stats is not None:
                    stats["slow_vars"] += int(slow.sum())
            if stats is not None:
                stats["vars"] += slow.size
            X[a] = Xn
        out = llr + np.add.reduceat(np.log(rho[:, order_v]), col_start, axis=1)
        # sign of the decision: the ratio state (what the syndrome test used), exact zero -> bit 0
        dec = (X < 1.0).astype(np.int8)
    return dec, out, iters


def main():
    import oracle
    from helpers import golden, ldpc_params, spa_contract
    p = ldpc_params("n1944")
    ec, ev = oracle.ldpc_edges(p)
    n_c, n_v = int(p["n_cnodes"]), int(p["n_vnodes"])
    g = golden("ldpc_c4y")
    ok = True
    for t in ("e8", "e9", "e10"):
        llr = g[t + "__llr"]
        st = dict(rows=0, near=0, blocks_it=0, blocks_near=0, slow_vars=0, vars=0)
        dec, out, its = decode(llr.copy(), ec, ev, n_c, n_v, int(g["iters"]), st)
        _, _, io = oracle.ldpc_bp_decode(llr.reshape(-1).copy(), p, "SPA", int(g["iters"]), True)
        same_dec = np.array_equal(dec, g[t + "__dec"])
        same_it = np.array_equal(its, io)
        try:
            spa_contract(out, g[t + "__out"], "ratio " + t)
            c = "inside the contract"
        except AssertionError as ex:
            c = "OUTSIDE: %s" % (ex,)
            ok = False
        dev = np.abs(out - g[t + "__out"])
        m = np.abs(g[t + "__out"]) < 26
        print("%s: dec_word equal %s, iterations equal %s (mean %.2f), %s; max dev below |LLR| 26: %.2e; near rows %.4f, block-iterations "
              "with a near row %.3f, slow variables %.2e" % (t, same_dec, same_it, its.mean(), c, dev[m].max(), st["near"] / st["rows"],
                                                              st["blocks_near"] / st["blocks_it"], st["slow_vars"] / max(st["vars"], 1)))
        ok = ok and same_dec and same_it
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
