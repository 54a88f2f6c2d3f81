#!/usr/bin/env python3
"""Randomised differential run against the CPU oracle (not part of the test-suite: minutes of GPU time, open-ended seeds).

    python scripts/fuzz_gpu.py [--seconds 120] [--seed 0]

Draws cases until the time is up: Viterbi (hard / soft / unquantized, random batch, block length, traceback depth, kernel path,
+-inf / 0 values), LDPC min-sum (random Tanner graphs, both decoder paths, special values; exact equality) and LDPC
sum-product (dec_word / iterations equal, LLRs within the suite's criterion), MAP decoding (4- and 8-state RSC, <= 1e-5),
turbo decoding (decoded bits equal except where the final LLR is ~0), PSK / QAM demodulation (hard: equal; soft: <= 1e-5) and -- round 6 --
the fused link front end against the staged kernels (bit patterns equal) and the oracle's demodulator.
Prints one line per failing case and a summary; exit status 1 if anything failed."""
import argparse
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from helpers import make_trellis  # noqa: E402
from test_random_codes_gpu import _random_ldpc  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args(argv)
    from commpy_amd import _lib
    from commpy_amd.channelcoding import RandInterlv, conv_encode_batch, ldpc_bp_decode, map_decode, turbo_decode, viterbi_decode
    from commpy_amd.modulation import PSKModem, QAMModem
    rs = np.random.RandomState(a.seed)
    tr7 = make_trellis("k7_133_171")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rsc = [make_trellis("rsc_legacy_4"), make_trellis("rsc_legacy_8")]
    others = {}
    t_end = time.time() + a.seconds
    n = {"viterbi": 0, "ldpc": 0, "map": 0, "turbo": 0, "demod": 0, "general": 0, "link": 0}
    modems = [QAMModem(4), QAMModem(16), QAMModem(64), QAMModem(256), PSKModem(2), PSKModem(4), PSKModem(8), PSKModem(16)]
    bad = []
    while time.time() < t_end:
        kind = rs.choice(["viterbi", "viterbi", "ldpc", "map", "turbo", "demod", "general", "link"])
        n[kind] += 1
        try:
            if kind == "viterbi" and rs.rand() < 0.35:
                # any trellis of the fixture list (k = 2, recursive, 8 .. 128 states: the state-per-lane / wide kernels), decoding
                # random received values -- no valid codeword needed to compare two decoders
                name = str(rs.choice(["t57", "rsc_legacy_4", "k2_default", "k2_lsb", "k2_rsc_matrix", "wifi_decimal_133_171",
                                      "rsc_legacy_8", "rsc_matrix_4", "r13_k4", "k5_23_35", "k8_247_371"]))
                if name not in others:
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        others[name] = make_trellis(name)
                tr = others[name]
                dtype = str(rs.choice(["hard", "soft", "unquantized"]))
                B, steps = int(rs.choice([1, 3, 16, 17, 64, 100])), int(rs.randint(tr.total_memory + 1, 120))
                length = steps * tr.n
                L = int(length * tr.k / tr.n)
                tb = None if rs.rand() < 0.4 else int(rs.randint(2, min(40, L) + 1))
                n_steps = int((L + tr.total_memory) / tr.k) - 1
                if (tb if tb is not None else min(5 * tr.total_memory, L)) - 1 > n_steps:
                    n[kind] -= 1                                   # no traceback ever runs: the reference returns uninitialised memory
                    continue
                if dtype == "hard":
                    rx = rs.randint(0, 2, (B, length)).astype(float)
                elif dtype == "soft":
                    rx = rs.randn(B, length) * rs.choice([0.5, 3.0])
                    rx[rs.rand(*rx.shape) < 0.01] = np.inf
                    if rs.rand() < 0.3:                            # round 3: NaN poisons the codeword like in the reference (detect and redo)
                        rx[rs.rand(*rx.shape) < 0.003] = np.nan
                else:
                    rx = rs.choice([-1.0, 1.0], size=(B, length)) + rs.randn(B, length) * 0.7
                want = oracle.viterbi_decode(rx, tr, tb, dtype)
                got = viterbi_decode(rx, tr, tb, dtype)
                if not np.array_equal(got, want):
                    bad.append(("viterbi-other", name, dtype, B, steps, tb, int(np.sum(got != want))))
                _lib.viterbi_set_path("general")                   # round 4: the general kernel on the same case
                got = viterbi_decode(rx, tr, tb, dtype)
                _lib.viterbi_set_path(None)
                if not np.array_equal(got, want):
                    bad.append(("viterbi-other-general", name, dtype, B, steps, tb, int(np.sum(got != want))))
                if name in ("t57", "k5_23_35"):                    # round 3: the small-ring fused kernel where it applies ("cw": fall back otherwise)
                    _lib.viterbi_set_path("cw")
                    got = viterbi_decode(rx, tr, tb, dtype)
                    _lib.viterbi_set_path(None)
                    if not np.array_equal(got, want):
                        bad.append(("viterbi-small-cw", name, dtype, B, steps, tb, _lib.last_kernel(), int(np.sum(got != want))))
            elif kind == "link":
                # round 6: the fused link front end (one launch: bits, conv_encode, puncturing, modulate, AWGN, soft demod, depuncturing)
                # against the seven staged kernels on the same counter-based streams -- message bits, noisy symbols, LLRs (zeros at the
                # punctured positions included) and error counts bit for bit; the LLRs against the oracle's demodulator on those symbols
                from commpy_amd.devicelink import DeviceWifiLink
                mcs = int(rs.choice([3, 4, 5, 6, 7, 8, 9]))
                gens = None if rs.rand() < 0.3 else [[0o133, 0o171]]
                chunk, agg = int(rs.choice([24, 120, 600, 1200, 2400])), int(rs.choice([1, 1, 2, 3]))
                T, snr, seed = int(rs.choice([1, 2, 7, 33, 64, 65, 200])), float(rs.uniform(-2.0, 45.0)), int(rs.randint(1, 1 << 30))
                got = {}
                try:
                    DeviceWifiLink(mcs, chunk, frame_aggregation=agg, generator_matrix=gens, seed=seed, fused=False)
                except ValueError:                                  # this chunk is not a whole number of symbols for this MCS
                    n[kind] -= 1
                    continue
                for fused in (True, False):
                    link = DeviceWifiLink(mcs, chunk, frame_aggregation=agg, generator_matrix=gens, seed=seed, fused=fused)
                    link.keep_rx = True
                    link._calls = int(seed % 5)                    # different stream ids
                    errs = link.run_batch(snr, T)
                    b = link._bufs
                    llr = (b['llr_de'] if link.keep_idx is not None else b['llr']).to_array((T, link.nde), np.float64)
                    got[fused] = (errs, b['msg'].to_array((T, link.nbits), np.uint8), b['rx'].to_array((T, link.nsym), np.complex128), llr)
                f_, s_ = got[True], got[False]
                same = (np.array_equal(f_[0], s_[0]) and np.array_equal(f_[1], s_[1]) and np.array_equal(f_[2].view(np.uint64), s_[2].view(np.uint64))
                        and np.array_equal(f_[3].view(np.uint64), s_[3].view(np.uint64)))
                nv = 2.0 * link.modem.Es / (link.rate * 10 ** (snr / 10.0))
                want = oracle.demodulate(link.modem.constellation, f_[2][:2].reshape(-1), "soft", nv).reshape(min(T, 2), -1)
                full = np.zeros((want.shape[0], link.nde))
                if link.de_idx is not None:
                    keep = link.de_idx >= 0
                    full[:, keep] = want[:, link.de_idx[keep]]
                else:
                    full = want
                fin = np.isfinite(full)
                ok = np.array_equal(fin, np.isfinite(f_[3][:2])) and (not fin.any() or np.max(np.abs(full[fin] - f_[3][:2][fin])) < 1e-5)
                if not same or not ok:
                    bad.append(("link", mcs, gens is None, chunk, agg, T, snr, seed, same, ok))
            elif kind == "viterbi" and rs.rand() < 0.25:
                # round 3: a random 64-state pair with both end taps through the table-driven fused kernel (default depth) or, at other
                # depths, whatever the forced-but-not-strict codeword path falls back to
                from commpy_amd.channelcoding import Trellis
                mem = int(rs.randint(2, 7))
                ends = (1 << mem) | 1
                g0, g1 = (int(ends | (rs.randint(0, 1 << (mem - 1)) << 1)) for _ in range(2))
                if g0 == g1:
                    g1 ^= 2
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    trg = Trellis(np.array([mem]), np.array([[g0, g1]]))
                dtype = str(rs.choice(["hard", "soft", "unquantized"]))
                B, nbits = int(rs.choice([1, 7, 64, 65, 130])), int(rs.randint(30, 300))
                tb = None if rs.rand() < 0.7 else int(rs.randint(2, min(49, nbits) + 1))
                coded = conv_encode_batch(rs.randint(0, 2, (B, nbits)), trg).astype(float)
                if dtype == "hard":
                    rx = np.where(rs.rand(*coded.shape) < 0.1, 1 - coded, coded)
                elif dtype == "soft":
                    rx = 4.0 * coded - 2 + rs.randn(*coded.shape) * rs.choice([0.5, 2.0])
                    if rs.rand() < 0.3:
                        rx[rs.rand(*rx.shape) < 0.002] = np.nan
                else:
                    rx = 2.0 * coded - 1 + rs.randn(*coded.shape) * rs.choice([0.3, 0.8])
                want = oracle.viterbi_decode(rx, trg, tb, dtype)
                for path in ("cw", None):
                    _lib.viterbi_set_path(path)
                    got = viterbi_decode(rx, trg, tb, dtype)
                    if not np.array_equal(got, want):
                        bad.append(("viterbi-table", oct(g0), oct(g1), dtype, B, nbits, tb, path, _lib.last_kernel(), int(np.sum(got != want))))
                _lib.viterbi_set_path(None)
            elif kind == "viterbi":
                dtype = rs.choice(["hard", "soft", "unquantized"])
                B, nbits = int(rs.choice([1, 2, 7, 63, 64, 65, 130, 300])), int(rs.randint(1, 400))
                # tb_depth - 1 > number of steps: the reference never runs a traceback and returns uninitialised memory
                # (convcode.py:711, :644) -- nothing to compare with; the largest defined depth is L + total_memory
                tb = None if rs.rand() < 0.4 else int(rs.randint(2, min(49, nbits + 6 + 6) + 1))
                coded = conv_encode_batch(rs.randint(0, 2, (B, nbits)), tr7).astype(float)
                if dtype == "hard":
                    rx = np.where(rs.rand(*coded.shape) < 0.1, 1 - coded, coded)
                elif dtype == "soft":
                    rx = 4.0 * coded - 2 + rs.randn(*coded.shape) * rs.choice([0.5, 2.0, 5.0])
                    for v in (np.inf, -np.inf, 0.0, 600.0):
                        rx[rs.rand(*rx.shape) < 0.005] = v
                    if rs.rand() < 0.3:
                        rx[rs.rand(*rx.shape) < 0.002] = np.nan
                else:
                    rx = 2.0 * coded - 1 + rs.randn(*coded.shape) * rs.choice([0.3, 0.8, 2.0])
                want = oracle.viterbi_decode(rx, tr7, tb, dtype)
                for path in ("cw!", "cw2!", "wave", "general", None):      # round 4: + the general kernel (viterbi_generic.hip)
                    _lib.viterbi_set_path(path)
                    got = viterbi_decode(rx, tr7, tb, dtype)
                    if not np.array_equal(got, want):
                        bad.append(("viterbi", dtype, B, nbits, tb, path, int(np.sum(got != want))))
                _lib.viterbi_set_path(None)
            elif kind == "general":
                # round 4: the argument domain beyond the specialised kernels -- random trellises with many states / k = 3 / n up to 12
                # (general Viterbi kernel; K = 9: the wide kernel), MAP above 16 states, checks of more than 32 edges, constellations
                # above 256 points
                from commpy_amd.channelcoding import Trellis
                from commpy_amd.modulation import Modem
                sub = int(rs.randint(5))
                if sub <= 1:
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        if sub == 0:
                            m = int(rs.choice([7, 8, 8, 9, 10]))
                            nn = int(rs.randint(2, 4))
                            gm = rs.randint(1, 2 ** (m + 1), (1, nn))
                            gm[0, 0] |= 1 | (1 << m)
                            trg = Trellis(np.array([m]), gm)
                        else:
                            mem = rs.randint(1, 3, 3)
                            nn = int(rs.randint(4, 13))
                            gm = rs.randint(0, 2 ** (int(mem.max()) + 1), (3, nn))
                            for i3 in range(3):
                                gm[i3, i3] |= 1
                            trg = Trellis(mem, gm)
                    try:
                        trg._device_handle()
                    except ValueError:
                        n[kind] -= 1
                        continue
                    dtype = str(rs.choice(["hard", "soft", "unquantized"]))
                    B, nbits = int(rs.choice([1, 2, 9, 65])), int(rs.randint(20, 160))
                    nbits -= nbits % trg.k
                    nbits = max(nbits, 6 * trg.k)
                    coded = conv_encode_batch(rs.randint(0, 2, (B, nbits)), trg).astype(float)
                    L = int(coded.shape[1] * trg.k / trg.n)
                    T = int((L + trg.total_memory) / trg.k) - 1
                    tb = None if rs.rand() < 0.5 else int(rs.randint(2, max(3, min(T, 70)) + 1))
                    if (tb if tb is not None else min(5 * trg.total_memory, L)) - 1 > T:
                        tb = max(2, T // 2)
                    if dtype == "hard":
                        rx = np.where(rs.rand(*coded.shape) < 0.1, 1 - coded, coded)
                    elif dtype == "soft":
                        rx = 4.0 * coded - 2 + rs.randn(*coded.shape) * rs.choice([1.0, 3.0])
                        rx[rs.rand(*rx.shape) < 0.004] = np.inf
                        if rs.rand() < 0.3:
                            rx[rs.rand(*rx.shape) < 0.003] = np.nan
                    else:
                        rx = 2.0 * coded - 1 + rs.randn(*coded.shape) * rs.choice([0.5, 1.2])
                    want = oracle.viterbi_decode(rx, trg, tb, dtype)
                    nvld = min(L, T * trg.k)
                    for path in (None, "general"):
                        _lib.viterbi_set_path(path)
                        got = viterbi_decode(rx, trg, tb, dtype)
                        if not np.array_equal(got[:, :nvld], want[:, :nvld]):
                            bad.append(("general-viterbi", trg.number_states, trg.k, trg.n, dtype, B, nbits, tb, path, _lib.last_kernel(),
                                        int(np.sum(got[:, :nvld] != want[:, :nvld]))))
                    _lib.viterbi_set_path(None)
                elif sub == 2:
                    m = int(rs.choice([5, 6, 7]))
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        trg = Trellis(np.array([m]), np.array([[1, int(rs.randint(1, 2 ** (m + 1))) | 1]]),
                                      np.array([[int(rs.randint(1, 2 ** (m + 1))) | 1 | (1 << m)]]), 'rsc')
                    B, N = int(rs.choice([1, 5, 70])), int(rs.randint(4, 60))
                    nv = float(rs.choice([0.4, 1.0]))
                    s_, p_, L_ = rs.randn(B, N) * 1.3, rs.randn(B, N) * 1.3, rs.randn(B, N) * rs.choice([0.0, 2.0])
                    Lx, bits = map_decode(s_, p_, trg, nv, L_, "decode")
                    for b in (0, B - 1):
                        Lo, bo = oracle.map_decode(s_[b], p_[b], trg, nv, L_[b], "decode")
                        if not (np.max(np.abs(Lx[b] - Lo)) < 1e-5) or np.any((bits[b] != bo) & (np.abs(Lo) > 1e-5)):
                            bad.append(("general-map", trg.number_states, B, N, nv))
                elif sub == 3:
                    n_c = int(rs.randint(3, 12))
                    n_v = int(rs.randint(60, 160))
                    deg = rs.randint(33, min(n_v - 1, 60) + 1, size=n_c)
                    p = _random_ldpc(rs, n_v, n_c, deg)
                    B, iters = int(rs.choice([1, 4, 33])), int(rs.randint(1, 6))
                    llr = rs.randn(B * n_v) * rs.choice([1.0, 3.0]) + rs.choice([0.5, 2.0])
                    for alg in ("MSA", "SPA"):
                        if alg == "SPA" and (np.max(np.abs(llr)) > 12.0 or iters > 3):
                            continue
                        do, oo, io = oracle.ldpc_bp_decode(llr.copy(), dict(p), alg, iters, True)
                        d, o, it = ldpc_bp_decode(llr.copy(), dict(p), alg, iters, return_iterations=True)
                        ok = np.array_equal(it, io) and "ldpc_exact_kernel" in _lib.last_kernel()
                        if alg == "MSA":
                            ok = ok and np.array_equal(o, oo) and np.array_equal(d, do)
                        else:
                            ok = ok and np.mean(np.abs(o - oo) <= 1e-5 + 1e-6 * np.abs(oo)) > 0.999
                        if not ok:
                            bad.append(("general-ldpc", alg, n_v, n_c, B, iters, _lib.last_kernel()))
                else:
                    nb = int(rs.choice([9, 10]))
                    M = 1 << nb
                    md = Modem((rs.randn(M) + 1j * rs.randn(M)) * 2.0) if sub == 3 or rs.rand() < 0.5 else QAMModem(1024)
                    ns = int(rs.choice([1, 5, 40]))
                    N0 = float(rs.choice([0.01, 0.3, 2.0]))
                    y = md.constellation[rs.randint(0, md.m, ns)] + np.sqrt(N0 / 2) * (rs.randn(ns) + 1j * rs.randn(ns))
                    hard, soft = md.demodulate(y, "hard"), md.demodulate(y, "soft", N0)
                    ho, so = oracle.demodulate(md.constellation, y, "hard"), oracle.demodulate(md.constellation, y, "soft", N0)
                    fin = np.isfinite(so)
                    if not np.array_equal(hard, ho) or not np.array_equal(np.isfinite(soft), fin) or \
                            not np.array_equal(soft[~fin], so[~fin], equal_nan=True) or \
                            (fin.any() and np.max(np.abs(soft[fin] - so[fin])) > 1e-5):
                        bad.append(("general-demod", md.m, ns, N0))
            elif kind == "ldpc":
                n_c = int(rs.randint(8, 120))
                n_v = int(n_c + rs.randint(8, 200))
                hi = int(min(31, n_v - 1, rs.randint(3, 32)))
                lo = int(rs.randint(2, hi + 1))
                p = _random_ldpc(rs, n_v, n_c, rs.randint(lo, hi + 1, size=n_c))
                B = int(rs.choice([1, 3, 64, 65, 200]))
                llr = rs.randn(B * n_v) * rs.choice([1.0, 3.0, 8.0]) + rs.choice([0.0, 1.5, 4.0])
                llr[rs.randint(0, llr.size, 6)] = 0.0
                if rs.rand() < 0.5:
                    llr[rs.randint(0, llr.size, 3)] = 1e4
                iters = int(rs.randint(1, 12))
                has_nan = rs.rand() < 0.25
                if has_nan:
                    llr[rs.randint(0, llr.size, 2)] = np.nan           # np.nan: one NaN sign (mixed signs are unspecified in NumPy's min)
                for alg in ("MSA", "SPA"):
                    if alg == "SPA":
                        if has_nan:
                            continue
                        # sum-product is only comparable where it is well conditioned: 2 atanh(x) amplifies a last-ulp
                        # difference of x by 1 / (1 - |x|), and BP on a random (dense, cyclic) graph amplifies that again
                        # every iteration -- measured on such graphs: 1e-15 after one iteration, 1e-2 after nine at |LLR| ~ 10,
                        # and a +-500 / 37 flip after ONE iteration at |LLR| ~ 36 (LDPC-SPA note).  Min-sum above is
                        # compared exactly, at every scale and iteration count.
                        if np.max(np.abs(llr)) > 12.0 or iters > 3:
                            continue
                    do, oo, io = oracle.ldpc_bp_decode(llr.copy(), dict(p), alg, iters, True)
                    for path in ("resident", "tiled"):
                        _lib.ldpc_set_path(path)
                        x = llr.copy()
                        try:
                            d, o, it = ldpc_bp_decode(x, dict(p), alg, iters, return_iterations=True)
                        except ValueError as exc:
                            # round 5: a FORCED path is never substituted -- the graph generator can push a check beyond 32 edges
                            # (it connects every variable), which only the literal kernel serves: that must be an error here, and
                            # the default dispatch must decode the case
                            if "only served by the literal kernel" not in str(exc):
                                raise
                            _lib.ldpc_set_path(None)
                            x = llr.copy()
                            d, o, it = ldpc_bp_decode(x, dict(p), alg, iters, return_iterations=True)
                            if "ldpc_exact_kernel" not in _lib.last_kernel():
                                bad.append(("ldpc-forced-path", alg, path, n_v, n_c, _lib.last_kernel()))
                        ok = np.array_equal(it, io) and np.nanmax(np.abs(x)) <= 500.0
                        if alg == "MSA":
                            ok = ok and np.array_equal(o, oo, equal_nan=True) and np.array_equal(d[~np.isnan(oo)], do[~np.isnan(oo)])
                        else:
                            fin = np.isfinite(oo)
                            ok = ok and np.array_equal(np.isfinite(o), fin)
                            if ok and fin.any():
                                dev, mag = np.abs(o[fin] - oo[fin]), np.abs(oo[fin])
                                ok = np.mean(dev <= 1e-5 + 1e-6 * mag) > 0.999
                        if not ok:
                            bad.append(("ldpc", alg, path, n_v, n_c, lo, hi, B, iters))
                _lib.ldpc_set_path(None)
            elif kind == "turbo":
                tr = rsc[int(rs.randint(2))]
                B, N, iters = int(rs.choice([1, 3, 16, 17, 40])), int(rs.randint(2, 200)), int(rs.randint(1, 5))
                il = RandInterlv(N, int(rs.randint(1 << 30)))
                nv = float(rs.choice([0.5, 1.0, 2.0]))
                rx = [rs.randn(B, N) * 1.2 + rs.choice([-1.0, 1.0], size=(B, N)) for _ in range(3)]
                got = turbo_decode(rx[0], rx[1], rx[2], tr, nv, iters, il)
                for b in range(min(B, 3)):
                    want = oracle.turbo_decode(rx[0][b], rx[1][b], rx[2][b], tr, nv, iters, il)
                    if np.mean(got[b] != want) > 0.02:             # a bit may differ only where the final LLR is within rounding of 0
                        bad.append(("turbo", tr.number_states, B, N, iters, nv, int(np.sum(got[b] != want))))
            elif kind == "demod":
                md = modems[int(rs.randint(len(modems)))]
                ns = int(rs.choice([1, 7, 256, 1000]))
                N0 = float(rs.choice([0.002, 0.05, 0.5, 2.0, 30.0]))    # round 4: + very high / very low SNR (closed-form and progression guards)
                y = md.constellation[rs.randint(0, md.m, ns)] + np.sqrt(N0 / 2) * (rs.randn(ns) + 1j * rs.randn(ns))
                if rs.rand() < 0.3:
                    y[rs.randint(0, ns)] *= 8.0                        # an outlier far outside the constellation
                hard, soft = md.demodulate(y, "hard"), md.demodulate(y, "soft", N0)
                ho, so = oracle.demodulate(md.constellation, y, "hard"), oracle.demodulate(md.constellation, y, "soft", N0)
                fin = np.isfinite(so)
                if not np.array_equal(hard, ho) or not np.array_equal(np.isfinite(soft), fin) or \
                        not np.array_equal(soft[~fin], so[~fin], equal_nan=True) or \
                        (fin.any() and np.max(np.abs(soft[fin] - so[fin])) > 1e-5):
                    bad.append(("demod", md.m, ns, N0))
            else:
                tr = rsc[int(rs.randint(2))]
                B, N = int(rs.choice([1, 5, 16, 17, 100])), int(rs.randint(1, 300))
                nv = float(rs.choice([0.3, 1.0, 3.0]))
                s_, p_ = rs.randn(B, N) * 1.5, rs.randn(B, N) * 1.5
                L = rs.randn(B, N) * rs.choice([0.0, 1.0, 4.0])
                if rs.rand() < 0.3:                                    # round 3: regimes where the reference's recursion underflows
                    amp, nv = float(rs.choice([5.0, 20.0])), float(rs.choice([0.02, 0.1, 1.0]))
                    s_, p_ = s_ * amp, p_ * amp
                    L = L * rs.choice([1.0, 15.0])
                L_ext, bits = map_decode(s_, p_, tr, nv, L, "decode")
                for b in range(min(B, 3)):
                    Lo, bo = oracle.map_decode(s_[b], p_[b], tr, nv, L[b], "decode")
                    fin = np.isfinite(Lo)
                    same = (np.array_equal(np.isnan(L_ext[b]), np.isnan(Lo)) and np.array_equal(np.isposinf(L_ext[b]), np.isposinf(Lo))
                            and np.array_equal(np.isneginf(L_ext[b]), np.isneginf(Lo)))
                    dev = float(np.max(np.abs(L_ext[b][fin] - Lo[fin]) - 1e-9 * np.abs(Lo[fin]))) if fin.any() else 0.0
                    if not same or dev > 1e-5 or np.any((bits[b] != bo) & ~(np.abs(Lo) <= 1e-5)):
                        bad.append(("map", tr.number_states, B, N, nv, same, dev))
        except Exception as e:                                     # noqa: BLE001
            if "check degree" in repr(e) and "not supported" in repr(e):
                n[kind] -= 1                                       # the generator attached orphans to a full check: engine limit (32), documented
            else:
                bad.append((kind, "exception", repr(e)[:200]))
            _lib.viterbi_set_path(None)
            _lib.ldpc_set_path(None)
    for b in bad[:40]:
        print("FAIL", b, flush=True)
    print("cases:", n, "failures:", len(bad), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
