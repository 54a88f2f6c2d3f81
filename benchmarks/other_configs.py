#!/usr/bin/env python3
"""The decoders north_star names next to the Viterbi headline, on their BASELINE.json configurations, as lines that the
DRIVER's one `bench.py` run carries (round 5: until now only the builder had ever timed them).

    run(lib, steps, warmup) -> list of dicts          (bench.py puts it into the JSON line as `other_configs`)
    python benchmarks/other_configs.py [--steps K]    (the same lines on their own, one JSON object per line)

Every entry is device-resident, timed with HIP events on the launch stream for the same number of steps as the headline,
names the kernel the library reports (cpx_last_kernel), carries algorithmic bytes per SURVEY 8(d) with the HBM fraction and
the resource that really binds, and ends with a parity sample of WHAT WAS TIMED against the CPU oracle (oracle/ is the
checker here, never the thing measured): decoded bits exact, min-sum LLRs <= 1e-5, sum-product under the banded contract
(tests/helpers.py::spa_contract), demodulator LLRs <= 1e-5.

  configs[2]  rate-1/3 turbo, 4-state RSC (1, 7/5), N = 1024, random interleaver, 6 iterations, Eb/N0 = 1.5 dB, B = 16 384
              (turbo.py:254-333)                                          25 600 B per codeword
  configs[3]  one GPU's share (B = 32 768 blocks) of the 802.11n (1944,1296) chain: 64-QAM symbols + AWGN at Eb/N0 = 8 dB ->
              soft demodulator (modulation.py:100-141) with the sign flip -> ldpc_bp_decode <= 50 iterations, min-sum and
              sum-product (ldpc.py:144-254)                               17 n B per block (decoder state resident)
  configs[4]  Wifi80211 MCS 5 (64-QAM, K = 7 r = 1/2 punctured to 2/3) end-to-end BER sweep Eb/N0 = 0 .. 10 dB, 1e8 information bits,
              everything device-resident (commpy_amd.devicelink.DeviceWifiLink; wifi80211.py:132-216, links.py:155-267); a step =
              the whole sweep; stage breakdown from HIP events between the stages (round 6)
  configs[0]  K = 3 (5,7) r = 1/2, 64-bit blocks, hard-decision Viterbi over a BSC(0.05) -- the reference's CPU plumbing case, here
              device-generated at B = 2^20 blocks (DeviceBscLink; test_convcode.py:133-178, channels.py:652-673) (round 6)
  pair        a K = 7 generator pair without a built-in instantiation, (135,147), on the configs[1] geometry: table-driven kernel and the
              pair's own code object (commpy_amd/jit.py, round 6)       17 510 B per codeword
  demod       the 64-QAM soft demodulator of that chain alone, (a) on the chain's 170 MB input, which FITS the 256 MiB
              Infinity Cache and is re-read every repetition, and (b) on rotating inputs of > 256 MiB in total, so that
              every byte comes from HBM                                   64 B per symbol
"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from commpy_amd import _lib  # noqa: E402

HBM_PEAK = 8000.0
DESIGN_1944 = os.path.join(ROOT, "commpy_amd", "channelcoding", "designs", "ldpc", "ieee80211n", "1944.1296.txt")


class Dev:
    """Tiny device-buffer helper on top of the C-ABI."""

    def __init__(self, lib):
        self.lib = lib
        self.bufs = []

    def put(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.empty(arr.nbytes)
        _lib.check(self.lib.cpx_memcpy_h2d(p, _lib.ptr(arr), arr.nbytes))
        return p

    def empty(self, nbytes):
        p = ctypes.c_void_p()
        _lib.check(self.lib.cpx_malloc(ctypes.byref(p), max(int(nbytes), 1)))
        self.bufs.append(p)
        return p

    def get(self, p, shape, dtype, offset=0):
        out = np.empty(shape, dtype=dtype)
        _lib.check(self.lib.cpx_memcpy_d2h(_lib.ptr(out), ctypes.c_void_p(p.value + offset), out.nbytes))
        return out

    def free(self):
        for p in self.bufs:
            self.lib.cpx_free(p)
        self.bufs = []


class Timers:
    """`n` HIP-event pairs (cpx_timer_*), recorded on the library's stream, read after the timed region."""

    def __init__(self, lib, n):
        self.lib = lib
        self.t = []
        for _ in range(n):
            tm = ctypes.c_void_p()
            _lib.check(lib.cpx_timer_create(ctypes.byref(tm)))
            self.t.append(tm)

    def start(self, i):
        _lib.check(self.lib.cpx_timer_start(self.t[i], None))

    def stop(self, i):
        _lib.check(self.lib.cpx_timer_stop(self.t[i], None))

    def read(self):
        out = []
        for tm in self.t:
            v = ctypes.c_float()
            _lib.check(self.lib.cpx_timer_elapsed_ms(tm, ctypes.byref(v)))
            out.append(v.value)
        for tm in self.t:
            self.lib.cpx_timer_destroy(tm)
        self.t = []
        return np.asarray(out, dtype=np.float64)


def warm(lib, fn, warmup, min_busy_s=0.03):
    """`warmup` untimed calls, then more of them until the GPU has been busy for `min_busy_s`: a fresh process raises its clock over
    the first ~20 ms of work (bench.py's per-step times fall from 1.65 to 1.53 ms over a dozen launches), and W calls of a 0.25 ms
    kernel are not that -- round 4's builder-run lines timed 3 steps after 1 warm-up call and read 4.28 ms where this reads 3.9."""
    for _ in range(max(warmup, 1)):
        fn()
    _lib.check(lib.cpx_stream_sync(None))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < min_busy_s:
        for _ in range(4):
            fn()
        _lib.check(lib.cpx_stream_sync(None))


def time_steps(lib, fn, steps, warmup):
    """Warm-up (see warm), then `steps` calls of fn(), each inside its own event pair; per-step milliseconds."""
    warm(lib, fn, warmup)
    tm = Timers(lib, steps)
    for i in range(steps):
        tm.start(i)
        fn()
        tm.stop(i)
    _lib.check(lib.cpx_stream_sync(None))
    return tm.read()


PMC_DIR = os.path.join(ROOT, "profiles")
PMC_ROUNDS = ("r06",)                                               # newest first; files <round>_oc_<name>_pmc.json


def pmc_fields(name, mix):
    """Counter figures for one entry from the committed rocprofv3 passes over THIS script (scripts/collect_pmc.py -- --which <name>;
    round-5 verdict item 4): `traffic` = HBM bytes of one step = sum over the step's kernels of launches x (FETCH_SIZE x gfx950
    correction + WRITE_SIZE) per launch, `valu.busy_frac` of the kernel a step spends most of its time in.  `mix` = {kernel-name
    prefix: launches per step}.  Counters cannot be read inside an unprofiled run, so they are quoted only when the file was recorded
    by a library built from exactly the sources of the one loaded now (cpx_build_id 'full' digest); otherwise the fields are null."""
    import json
    none = {"traffic": None, "traffic_source": None, "valu": None}
    build = _lib.build_id()
    for rnd in PMC_ROUNDS:
        path = os.path.join(PMC_DIR, "%s_oc_%s_pmc.json" % (rnd, name))
        try:
            with open(path) as f:
                tj = json.load(f)
        except (OSError, ValueError):
            continue
        if not build.get("full") or (tj.get("build_id") or {}).get("full") != build.get("full"):
            continue
        traffic, per_kernel, dom, dom_ns = 0.0, {}, None, -1.0
        for prefix, launches in mix.items():
            hit = [(k, v) for k, v in tj.get("kernels", {}).items() if k.startswith(prefix) and "traffic_bytes_per_launch" in v]
            if not hit:
                return none
            k, v = max(hit, key=lambda kv: kv[1].get("duration_ns_avg", 0) * kv[1].get("calls", 1))
            traffic += launches * v["traffic_bytes_per_launch"]
            per_kernel[k] = {"launches_per_step": launches, "traffic_bytes_per_launch": v["traffic_bytes_per_launch"],
                             "duration_us_under_profiler": v.get("duration_ns_avg", 0) / 1e3,
                             "valu_busy_frac": (v.get("valu") or {}).get("busy_frac"),
                             "lds_bank_conflict_frac_of_lds_active": (v["SQ_LDS_BANK_CONFLICT"] / v["SQ_ACTIVE_INST_LDS"]
                                                                      if v.get("SQ_ACTIVE_INST_LDS") else None)}
            if launches * v.get("duration_ns_avg", 0) > dom_ns:
                dom, dom_ns = v, launches * v.get("duration_ns_avg", 0)
        return {"traffic": traffic, "valu": dom.get("valu"), "per_kernel": per_kernel,
                "traffic_source": "profiles/%s (rocprofv3 FETCH_SIZE x %.1f + WRITE_SIZE passes over benchmarks/other_configs.py "
                                  "--which %s; git %s, build id %s = the loaded library's)"
                                  % (os.path.basename(path), tj.get("fetch_scale", 2.0), name, str(tj.get("git_head"))[:12],
                                     build.get("full"))}
    return none


def entry(config, workload, kernel, ms, units, unit_name, alg_bytes, bound, parity, extra=None, dtype="f64", pmc=None):
    avg = float(np.mean(ms))
    ach = alg_bytes / (avg * 1e-3) / 1e9
    d = {"config": config, "workload": workload, "kernel": kernel, "ms": avg, "ms_min": float(np.min(ms)),
         "ms_median": float(np.median(ms)), "steps": int(len(ms)), "value": units / (avg * 1e-3), "unit": unit_name + "/s",
         "dtype": dtype,
         "roofline": {"bound": bound, "achieved": ach, "peak": HBM_PEAK, "unit": "GB/s", "frac": ach / HBM_PEAK,
                      "algorithmic_bytes_per_launch": int(alg_bytes)},
         "parity": parity}
    if pmc is not None:
        got = pmc_fields(*pmc)
        d["roofline"]["traffic"] = got["traffic"]
        d["roofline"]["traffic_source"] = got["traffic_source"]
        d["roofline"]["valu"] = got["valu"]
        if got.get("per_kernel"):
            d["roofline"]["per_kernel"] = got["per_kernel"]
    if extra:
        d.update(extra)
    return d


# ---------------------------------------------------------------------------------------------------------------------------
def turbo_workload(B=16384, N=1024, states=4):
    """SURVEY 8(d) C3 (seeds as in benchmarks/bench_kernels.py): B distinct codewords, encoded by the device encoder."""
    import warnings
    from commpy_amd.channelcoding import RandInterlv, Trellis
    from commpy_amd.devicelink import turbo_encode_gpu
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        # 4 states: BASELINE config 3;  8 states: the LTE / UMTS constituent code (1, 15/13) (SURVEY 8d "optional 8-state")
        tr = Trellis(np.array([2]), np.array([[1, 7]]), 5, "rsc") if states == 4 else Trellis(np.array([3]), np.array([[1, 0o15]]), 0o13, "rsc")
    il = RandInterlv(N, 1234)
    rs = np.random.RandomState(20)
    msgs = rs.randint(0, 2, (B, N))
    s, p1, p2 = (a[:, :N] * 2.0 - 1 for a in turbo_encode_gpu(msgs, tr, tr, il))
    nv = 1 / (2 * (1.0 / 3) * 10 ** (1.5 / 10.0))
    nrs = np.random.RandomState(21)
    s, p1, p2 = (a + np.sqrt(nv) * nrs.randn(B, N) for a in (s, p1, p2))
    return tr, il, msgs, s, p1, p2, nv


def run_turbo(lib, steps, warmup, B=16384, n_check=64):
    import oracle
    N, n_iter = 1024, 6
    tr, il, msgs, s, p1, p2, nv = turbo_workload(B, N)
    dev = Dev(lib)
    try:
        d_s, d_p1, d_p2 = dev.put(s), dev.put(p1), dev.put(p2)
        d_perm = dev.put(np.asarray(il.p_array, dtype=np.int32))
        d_bits = dev.empty(B * N)
        h = tr._device_handle()
        ms = time_steps(lib, lambda: _lib.check(lib.cpx_turbo_decode_batch_dev(h, d_s, d_p1, d_p2, None, d_perm, B, N, nv, n_iter,
                                                                              d_bits, None)), steps, warmup)
        kname = _lib.last_kernel()
        bits = dev.get(d_bits, (B, N), np.uint8)
    finally:
        dev.free()
    # parity of what was timed: the first, middle and last codewords (first / last workgroup, every lane group) against the oracle
    idx = sorted(set(list(range(min(n_check // 2, B))) + list(range(B // 2, min(B // 2 + n_check // 4, B))) +
                     list(range(max(B - n_check // 4, 0), B))))
    t0 = time.perf_counter()
    mism = sum(int(np.sum(oracle.turbo_decode(s[i], p1[i], p2[i], tr, nv, n_iter, il) != bits[i])) for i in idx)
    return entry("configs[2]", "rate-1/3 turbo, 4-state RSC (1,7/5), N=1024, random interleaver, 6 iterations, Eb/N0=1.5 dB, "
                 "B=%d" % B, kname, ms, B * N, "info-bits", B * 25600, "latency/hbm-traffic",
                 {"vs": "oracle turbo_decode (turbo.py:254-333)", "codewords": len(idx), "mismatched_bits": mism,
                  "ok": mism == 0, "oracle_s": round(time.perf_counter() - t0, 2)},
                 {"ber": float(np.mean(bits != msgs)),
                  "bytes_model": "SURVEY 8d: 3 x 8 N in + N out = 25 600 B per codeword; the decoder really moves ~40 x that "
                                 "between its passes (roofline.traffic), so this kernel is far from the HBM roofline "
                                 "by construction"},
                 pmc=("turbo", {"turbo_pass_kernel": 2 * n_iter, "turbo_init_kernel": 1, "turbo_final_kernel": 1}))


# ---------------------------------------------------------------------------------------------------------------------------
def _spa_contract(out, want):
    """tests/helpers.py::spa_contract as a verdict instead of an assertion: (ok, detail)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        from helpers import spa_contract
        spa_contract(out, want, "bench")
        return True, None
    except AssertionError as exc:
        return False, str(exc)[:200]
    finally:
        sys.path.pop(0)


def run_config4(lib, steps, warmup, B=32768, ebn0=8.0, n_check=24, big_rotation_bytes=320 << 20):
    """configs[3], one GPU's share: 64-QAM + AWGN (device generated) -> soft demod with sign flip -> LDPC BP (MSA, SPA).
    Returns the entries for the two chains and for the demodulator alone."""
    import oracle
    from commpy_amd.channelcoding.ldpc import _device_code, get_ldpc_code_params
    from commpy_amd.devicelink import LdpcEncoder
    from commpy_amd.modulation import QAMModem
    p = get_ldpc_code_params(DESIGN_1944, True)
    md = QAMModem(64)
    n, nsym, k = 1944, 324, 1296
    N0 = 42.0 / ((2.0 / 3) * 6 * 10 ** (ebn0 / 10.0))
    sc = float(np.sqrt(N0 / 2))
    out = []
    dev = Dev(lib)
    try:
        enc = LdpcEncoder(p, "gf2")
        d_msg, d_bits = dev.empty(B * enc.k), dev.empty(B * n)
        d_sym, d_y = dev.empty(B * nsym * 16), dev.empty(B * nsym * 16)
        d_neg = dev.empty(B * n * 8)
        d_dec, d_out, d_it = dev.empty(B * n), dev.empty(B * n * 8), dev.empty(B * 4)
        code, h_md = _device_code(p), md._device_handle()
        # untimed: the batch that every timed step then decodes (same seeds as benchmarks/bench_kernels.py config4)
        _lib.check(lib.cpx_random_bits_dev(d_msg, B * enc.k, 30, 0, None))
        enc.encode_dev(d_msg, B, d_bits)
        _lib.check(lib.cpx_modulate_dev(h_md, d_bits, B * nsym, d_sym, None))
        _lib.check(lib.cpx_awgn_dev(d_sym, B * nsym, sc, sc, 31, 1, d_y, None))
        _lib.check(lib.cpx_stream_sync(None))
        sent = dev.get(d_bits, (B, n), np.int8)
        ncheck = min(n_check, B)
        y_chk = dev.get(d_y, (ncheck * nsym,), np.complex128)
        llr_chk = -oracle.demodulate(md.constellation, y_chk, "soft", N0)          # what the decoder is fed (sign flip, quirk B6)

        for alg, name in ((1, "MSA"), (0, "SPA")):
            # one step = demodulator (writes the LLRs the decoder then clips in place, ldpc.py:186) + decoder; an event pair each
            def chain():
                _lib.check(lib.cpx_demod_soft_scaled_dev(h_md, d_y, B * nsym, float(N0), -1.0, d_neg, None))
                _lib.check(lib.cpx_ldpc_bp_decode_batch_bm_dev(code, d_neg, B, alg, 50, d_dec, d_out, d_it, None))

            warm(lib, chain, warmup)
            tm = Timers(lib, 2 * steps)
            for i in range(steps):
                tm.start(2 * i)
                _lib.check(lib.cpx_demod_soft_scaled_dev(h_md, d_y, B * nsym, float(N0), -1.0, d_neg, None))
                tm.stop(2 * i)
                tm.start(2 * i + 1)
                _lib.check(lib.cpx_ldpc_bp_decode_batch_bm_dev(code, d_neg, B, alg, 50, d_dec, d_out, d_it, None))
                tm.stop(2 * i + 1)
            _lib.check(lib.cpx_stream_sync(None))
            kname = _lib.last_kernel()
            t = tm.read()
            ms_demod, ms_dec = t[0::2], t[1::2]
            its = dev.get(d_it, (B,), np.int32)
            dec = dev.get(d_dec, (B, n), np.int8)
            o_chk = dev.get(d_out, (ncheck, n), np.float64)
            # parity, kernel by kernel on the SAME inputs: the LLRs the demodulator handed over (read back: the decoder has clipped them
            # in place, which the oracle's own clip repeats) against the oracle's demodulator, and the decoder's outputs against the
            # oracle's decoder fed with exactly those LLRs -- bits and iteration counts exact; LLRs: MSA 1e-5, SPA the banded contract.
            # (Feeding the oracle decoder the ORACLE's LLRs instead compares two decodes of inputs 3e-14 apart: at 8 dB min-sum leaves
            # most blocks unconverged and amplifies that to 5e-2 over 50 iterations.)
            t0 = time.perf_counter()
            llr_eng = dev.get(d_neg, (ncheck * n,), np.float64)
            demod_err = float(np.max(np.abs(llr_eng - np.clip(llr_chk, -500.0, 500.0))))
            want_dec, want_out, want_it = oracle.ldpc_bp_decode(llr_eng.copy(), p, name, 50, return_iters=True)
            want_dec, want_out = np.atleast_2d(want_dec.T), np.atleast_2d(want_out.T)     # [blocks][n]
            bits_ok = bool(np.array_equal(want_dec, dec[:ncheck]))
            its_ok = bool(np.array_equal(want_it, its[:ncheck]))
            if name == "MSA":
                fin = np.isfinite(want_out)
                dmax = float(np.max(np.abs(want_out[fin] - o_chk[fin]))) if fin.any() else 0.0
                llr_ok, detail = bool(dmax <= 1e-5 and np.array_equal(fin, np.isfinite(o_chk))), {"max_abs_llr_err": dmax, "tolerance": 1e-5}
            else:
                llr_ok, why = _spa_contract(o_chk, want_out)
                below = np.abs(want_out) < 10.0
                detail = {"contract": "tests/helpers.py::spa_contract (|LLR|<10: 1e-5; [10,26): 99.95 % within 1e-5, none beyond 2e-4; "
                                      ">=26: sign and finiteness)", "max_abs_llr_err_below_10": float(np.max(np.abs(want_out[below] - o_chk[below]))) if below.any() else 0.0,
                          "violation": why}
            parity = {"vs": "oracle ldpc_bp_decode on the LLRs the device demodulator produced (ldpc.py:144-254); those LLRs vs oracle demodulate "
                            "(modulation.py:100-141)", "blocks": int(ncheck),
                      "dec_word_equal": bits_ok, "iterations_equal": its_ok, "llr_ok": llr_ok, "demod_max_abs_err": demod_err,
                      "ok": bits_ok and its_ok and llr_ok and demod_err <= 1e-5,
                      "oracle_s": round(time.perf_counter() - t0, 2)}
            parity.update(detail)
            E = 7128
            out.append(entry("configs[3] (one GPU's share), decoder", "802.11n (1944,1296) r=2/3 ldpc_bp_decode %s, <=50 iterations, LLRs of the "
                             "64-QAM chain at Eb/N0=%.0f dB, B=%d blocks, mean executed iterations %.2f" % (name, ebn0, B, its.mean()),
                             kname, ms_dec, B * k, "info-bits", B * n * 17, "valu+lds" if name == "MSA" else "valu", parity,
                             {"algorithm": name, "mean_iterations": float(its.mean()), "max_iterations": int(its.max()),
                              "block_iterations_per_s": float(its.sum()) / (float(np.mean(ms_dec)) * 1e-3),
                              "frame_error_rate": float(np.mean((dec != sent).any(axis=1))),
                              "bit_error_rate": float(np.mean(dec[:, :k] != sent[:, :k])),
                              "chain_ms": float(np.mean(ms_demod + ms_dec)),
                              "chain_info_bits_per_s": B * k / (float(np.mean(ms_demod + ms_dec)) * 1e-3),
                              "bytes_model": "SURVEY 8d resident model: 17 n B per block (llr in, out_llrs + dec_word out), whatever the "
                                             "iteration count; the decoder state lives in LDS",
                              "survey_8d_streaming_formulation_bytes_per_launch": int(its.sum()) * (4 * E + 2 * n) * 8 + B * n * 17},
                             pmc=("config4", {"ldpc_resident_kernel<": 1} if name == "MSA" else {"ldpc_resident_ratio_kernel": 1})))
            if name == "MSA":
                ms_demod_chain = ms_demod

        # ---- the demodulator alone: (a) the chain's own input (fits the Infinity Cache), (b) rotating inputs > 256 MiB ----
        ns = B * nsym
        d_llr = d_neg
        ms_a = time_steps(lib, lambda: _lib.check(lib.cpx_demod_soft_dev(h_md, d_y, ns, float(N0), d_llr, None)), steps, warmup)
        kname = _lib.last_kernel()
        got = dev.get(d_llr, (ncheck * n,), np.float64)
        dmax = float(np.max(np.abs(got + llr_chk)))
        par = {"vs": "oracle demodulate (modulation.py:100-141)", "symbols": int(ncheck * nsym), "max_abs_llr_err": dmax, "tolerance": 1e-5,
               "ok": bool(dmax <= 1e-5)}
        in_bytes = ns * 16
        nrot = max(2, -(-big_rotation_bytes // in_bytes))
        rot = [d_y] + [dev.empty(in_bytes) for _ in range(nrot - 1)]
        for j, d in enumerate(rot[1:]):
            _lib.check(lib.cpx_awgn_dev(d_sym, ns, sc, sc, 131 + j, 1, d, None))
        outs = [d_llr, d_out]                                                   # two output arrays in turn (each ns * 48 B)
        cnt = [0]

        def rot_step():
            i = cnt[0]
            cnt[0] += 1
            _lib.check(lib.cpx_demod_soft_dev(h_md, rot[i % nrot], ns, float(N0), outs[i % 2], None))

        ms_b = time_steps(lib, rot_step, max(steps, nrot), max(warmup, nrot))
        for tag, ms, wl in (("cache-resident input", ms_a, "the chain's %d MB input re-read every repetition: it fits the 256 MiB "
                             "Infinity Cache, FETCH_SIZE counts its hits" % (in_bytes >> 20)),
                            ("HBM-resident input", ms_b, "%d inputs of %d MB used in turn (%d MB > 256 MiB Infinity Cache): every byte "
                             "comes from HBM -- THIS is the HBM fraction" % (nrot, in_bytes >> 20, (nrot * in_bytes) >> 20))):
            out.append(entry("64-QAM soft demodulator, " + tag, "QAMModem(64).demodulate(y, 'soft', N0), %d symbols, Es/N0 of the "
                             "config-3 chain; %s" % (ns, wl), kname, ms, ns, "symbols", ns * 64, "valu (exp/log) + hbm", par,
                             {"bytes_model": "SURVEY 8d: 16 + 8 x 6 = 64 B per symbol", "chain_demod_ms": float(np.mean(ms_demod_chain))},
                             pmc=("config4", {"demod_soft_sep_kernel<3": 1})))
    finally:
        dev.free()
    return out


# ---------------------------------------------------------------------------------------------------------------------------
def run_config5(lib, steps, warmup, total_bits=1e8, n_check=48):
    """configs[4]: one step = the whole 11-point sweep.  Parity of what was timed: the decoder's bits for sampled frames of the LAST
    timed sweep against the oracle on the very LLRs the device chain produced, and those LLRs (last SNR point: its symbols are still
    in HBM) against the oracle's demodulator + depuncturing."""
    import math
    import oracle
    from commpy_amd.devicelink import DeviceWifiLink
    link = DeviceWifiLink(5, 1200, generator_matrix=[[0o133, 0o171]], seed=2026)
    ebn0 = np.arange(0.0, 10.5, 1.0)
    snrs = ebn0 + 10 * math.log10(link.rate * link.modem.num_bits_symbol)
    per_point = int(total_bits / len(snrs))
    marks = ("front end (bits, conv_encode, puncture, 64-QAM, AWGN, soft demod, depuncture: %s) x 11 points"
             % ("one fused launch per point" if link._front is not None else "seven kernels per point"), "viterbi_decode, all frames",
             "error count")
    state = {"bers": None, "tm": None, "i": 0}

    def sweep():
        state["bers"] = link.ber_sweep_batched(snrs, per_point, mark=state["mark"])

    state["mark"] = None
    warm(lib, sweep, warmup, min_busy_s=0.05)
    tm_all = Timers(lib, steps)
    tm_stage = Timers(lib, 3 * steps)

    def mark_fn(k, start):                                          # k-th stage of step state["i"]
        j = 3 * state["i"] + k
        (tm_stage.start if start else tm_stage.stop)(j)
    state["mark"] = mark_fn
    for i in range(steps):
        state["i"] = i
        tm_all.start(i)
        sweep()
        tm_all.stop(i)
    _lib.check(lib.cpx_stream_sync(None))
    ms = tm_all.read()
    st = tm_stage.read().reshape(steps, 3)
    kname = _lib.last_kernel()
    bers = state["bers"]
    # ---- parity of the last timed sweep ----
    bufs = link._bufs
    T = int(math.ceil(per_point / link.nbits))
    R = len(snrs) * T
    rs = np.random.RandomState(3)
    rows = sorted(set(rs.randint(0, R, n_check).tolist()) | {0, R - 1, 65535, min(65536, R - 1)})
    t0 = time.perf_counter()
    mism = 0
    for r in rows:
        llr = np.empty(link.nde)
        _lib.check(lib.cpx_memcpy_d2h(_lib.ptr(llr), ctypes.c_void_p(bufs['llr_all'].ptr.value + r * link.nde * 8), llr.nbytes))
        dec = np.empty(link.nbits, np.uint8)
        _lib.check(lib.cpx_memcpy_d2h(_lib.ptr(dec), ctypes.c_void_p(bufs['dec'].ptr.value + r * link.nbits), dec.nbytes))
        mism += int(np.sum(oracle.viterbi_decode(llr, link.trellis, None, "soft")[:link.nbits] != dec))
    # demodulator + depuncturing of the last point, first frames
    nf = 4
    noise_std = math.sqrt(2.0 * link.modem.Es / (link.rate * 10 ** (float(snrs[-1]) / 10.0)))
    fused = link._front is not None and "link_front" in link.front_last_kernel
    if fused:
        # the fused launch never stores the symbols: the first frames of the last point once more from the same counter-based streams,
        # this time with them (front_sample) -- and the LLRs of that launch must be the sweep's own, bit for bit
        _, y2, llr2 = link.front_sample(nf, float(snrs[-1]), link._calls)
        y = y2.reshape(-1)
    else:
        y = np.empty(nf * link.nsym, np.complex128)
        _lib.check(lib.cpx_memcpy_d2h(_lib.ptr(y), bufs['sym'].ptr, y.nbytes))
    want = oracle.demodulate(link.modem.constellation, y, "soft", noise_std ** 2).reshape(nf, -1)
    got = np.empty((nf, link.nde))
    _lib.check(lib.cpx_memcpy_d2h(_lib.ptr(got), ctypes.c_void_p(bufs['llr_all'].ptr.value + (len(snrs) - 1) * T * link.nde * 8), got.nbytes))
    full = np.zeros((nf, link.nde))
    if link.de_idx is not None:
        keep = link.de_idx >= 0
        full[:, keep] = want[:, link.de_idx[keep]]
    else:
        full = want
    llr_err = float(np.max(np.abs(full - got)))
    if fused and not np.array_equal(llr2.view(np.uint64), got.view(np.uint64)):
        llr_err = float("inf")                                      # the regenerated launch is not the sweep's
    total = R * link.nbits
    front_bytes = (R * (link.nbits + link.nde * 8) if fused else
                   R * (link.nbits + link.ncoded + link.ntx + link.nsym * 16 * 3 + link.nsym * 6 * 8 * 2 + link.nde * 8))
    alg = R * (link.nde * 8 + link.nbits)                            # decoder model, SURVEY 8d: LLRs in, bits out
    return entry("configs[4] (one GPU)", "Wifi80211 MCS 5 (64-QAM, K=7 (133,171) r=1/2 punctured to 2/3), frames of %d info bits, "
                 "Eb/N0 = 0..10 dB in 11 points, %d frames = %.3g info bits per sweep; a step = the whole device-resident sweep"
                 % (link.nbits, R, total), kname + (" <- " + link.front_last_kernel.split(" (")[0] if fused else ""), ms, total,
                 "simulated info-bits", alg, "valu (Viterbi) + " + ("valu (front end)" if fused else "hbm (front end)"),
                 {"vs": "oracle viterbi_decode (convcode.py:661-749) on the LLRs of the last timed sweep; oracle demodulate "
                        "(modulation.py:100-141) + depuncturing (convcode.py:777-804) on the symbols of its last point",
                  "frames": len(rows), "mismatched_bits": mism, "demod_depuncture_max_abs_err": llr_err, "tolerance": 1e-5,
                  "ok": mism == 0 and llr_err < 1e-5, "oracle_s": round(time.perf_counter() - t0, 2)},
                 {"stage_ms": {m: float(np.mean(st[:, k])) for k, m in enumerate(marks)},
                  "ebn0_db": ebn0.tolist(), "ber": [float(b) for b in bers],
                  "bytes_model": "decoder only (SURVEY 8d): %d x 8 B of LLRs in + %d B of bits out per frame; the front end moves ~%d MB "
                                 "per sweep (%s)"
                                 % (link.nde, link.nbits, front_bytes >> 20, "fused launch: only the message bits and the decoder's "
                                    "LLRs are written" if fused else "messages, coded bits, symbols written / read / noised, LLRs "
                                    "written, gathered"),
                  "reference_cpu": "Wifi80211(5).link_performance: ~1e3 info-bits/s on one core (BASELINE.md)"},
                 pmc=("config5", {"viterbi_cw_fused_kernel": 1, "viterbi_wave_kernel": 1, "link_front_kernel": len(snrs),
                                  "count_errors_kernel": 1}))


def run_config1(lib, steps, warmup, B=1 << 20, n_check=3000):
    """configs[0] on the device: the timed step is the hard-decision decode of B blocks; messages, encoder and BSC are generated once."""
    import warnings
    import oracle
    from commpy_amd.channelcoding import Trellis
    from commpy_amd.devicelink import DeviceBscLink
    tr = Trellis(np.array([2]), np.array([[5, 7]]))
    link = DeviceBscLink(tr, 64, seed=2)
    bufs = link.generate(0.05, B)
    ms = time_steps(lib, lambda: link.decode(B), steps, warmup)
    kname = _lib.last_kernel()
    _lib.check(lib.cpx_stream_sync(None))
    idx = np.r_[0:n_check // 2, B - n_check // 2:B]
    rx = np.concatenate([bufs['rx'].to_array((B, link.ncoded), np.float64)[idx]])
    dec = bufs['dec'].to_array((B, link.L), np.uint8)
    msg = bufs['msg'].to_array((B, 64), np.uint8)
    t0 = time.perf_counter()
    mism = int(np.sum(oracle.viterbi_decode(rx, tr, None, "hard") != dec[idx]))
    return entry("configs[0], device-generated", "K=3 (5,7) r=1/2, 64-bit blocks ('term': 132 coded bits), hard-decision Viterbi over "
                 "BSC(0.05), tb_depth=10, B=%d blocks generated on the device" % B, kname, ms, B * 64, "info-bits",
                 B * (link.ncoded * 8 + link.L), "hbm + valu",
                 {"vs": "oracle viterbi_decode 'hard' (convcode.py:661-749) on the device-generated channel output", "codewords": len(idx),
                  "mismatched_bits": mism, "ok": mism == 0, "oracle_s": round(time.perf_counter() - t0, 2)},
                 pmc=("config1", {"viterbi_cw_fused_kernel": 1}), extra=
                 {"ber": float(np.mean(dec[:, :64] != msg)),
                  "bytes_model": "SURVEY 8d form: %d float64 received values in + %d decoded bytes out per block" % (link.ncoded, link.L)})


def run_pair(lib, steps, warmup, B=65536, n_check=192):
    """Round 6: a K = 7 pair that is NOT built in -- (135,147) -- on the headline geometry (1024-bit blocks, soft, tb_depth 30), through the
    table-driven fused kernel and through the pair's own code object (Trellis.specialize, commpy_amd/jit.py: compiled by hipcc when the
    on-disk cache does not have it).  Both against the oracle on sampled codewords, and against each other on the whole batch."""
    import oracle
    from commpy_amd import jit
    from commpy_amd.channelcoding import Trellis, conv_encode_batch
    tr = Trellis(np.array([6]), np.array([[0o135, 0o147]]))
    rs = np.random.RandomState(4)
    coded = conv_encode_batch(rs.randint(0, 2, (B, 1024)).astype(np.uint8), tr).astype(np.float64)
    llr = np.ascontiguousarray(4.0 * coded - 2 + rs.standard_normal(coded.shape).astype(np.float32) * 1.4, dtype=np.float64)
    del coded
    dev = Dev(lib)
    out = []
    try:
        d_in, d_out = dev.put(llr), dev.empty(B * 1030)
        h = tr._device_handle()
        idx = np.r_[0:n_check // 2, B - n_check // 2:B]
        want = oracle.viterbi_decode_mt(llr[idx], tr, None, "soft")
        bits = {}
        for what in ("table-driven kernel", "its own code object"):
            t0 = time.perf_counter()
            if what == "its own code object" and not tr.specialize():
                out.append({"config": "configs[1] geometry, pair (135,147), " + what, "skipped": "no code object: %s" % jit.viterbi_code_object.last_error})
                continue
            t_jit = time.perf_counter() - t0
            ms = time_steps(lib, lambda: _lib.check(lib.cpx_viterbi_decode_batch_dev(h, d_in, B, 2060, 1030, 1030, 30, 1, d_out, None)), steps, warmup)
            kname = _lib.last_kernel()
            bits[what] = dev.get(d_out, (B, 1030), np.uint8)
            mism = int(np.sum(bits[what][idx] != want))
            par = {"vs": "oracle viterbi_decode 'soft' (convcode.py:661-749)", "codewords": len(idx), "mismatched_bits": mism, "ok": mism == 0}
            if len(bits) == 2:
                par["bits_differing_from_the_table_driven_kernel_whole_batch"] = int(np.sum(bits["table-driven kernel"] != bits[what]))
                par["ok"] = par["ok"] and par["bits_differing_from_the_table_driven_kernel_whole_batch"] == 0
            out.append(entry("configs[1] geometry, pair (135,147), " + what, "K=7 (135,147) r=1/2 -- no built-in instantiation --, 1024-bit blocks, soft "
                             "Viterbi, tb_depth=30, B=%d" % B, kname, ms, B * 1024, "info-bits", B * 17510, "valu", par,
                             {"specialize_s": round(t_jit, 2)} if what != "table-driven kernel" else None,
                             pmc=("pair", {"viterbi_cw_fused_kernel<6, 0u, 0u": 1} if what == "table-driven kernel" else
                                  {"viterbi_cw_fused_kernel<6, 93u, 115u": 1})))
    finally:
        dev.free()
    return out


def run(lib, steps=20, warmup=3, scale=1.0, log=None, which=None):
    """All entries (or the ones named in `which`); never raises for a failing workload (the headline line must not be lost): a
    failure becomes an entry with `error`."""
    out = []
    for name, fn in (("turbo", lambda: [run_turbo(lib, steps, warmup, B=int(16384 * scale))]),
                     ("config4", lambda: run_config4(lib, steps, warmup, B=int(32768 * scale))),
                     ("config5", lambda: [run_config5(lib, steps, warmup, total_bits=1e8 * scale)]),
                     ("config1", lambda: [run_config1(lib, steps, warmup, B=int((1 << 20) * scale))]),
                     ("pair", lambda: run_pair(lib, steps, warmup, B=int(65536 * scale)))):
        if which and name not in which:
            continue
        t0 = time.perf_counter()
        try:
            got = fn()
        except Exception as exc:                                   # reported, not hidden
            got = [{"config": name, "error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}]
        for g in got:
            g["wall_s_incl_setup_and_oracle"] = round(time.perf_counter() - t0, 2)
        out.extend(got)
        if log:
            log("other_configs: %s done in %.1f s" % (name, time.perf_counter() - t0))
    return out


if __name__ == "__main__":
    import argparse
    import json
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--which", default="", help="comma list of turbo,config4,config5,config1,pair (default: all)")
    a = ap.parse_args()
    lib = _lib.load()
    _lib.require_device()
    for e in run(lib, a.steps, a.warmup, a.scale, which=set(a.which.split(",")) if a.which else None):
        print(json.dumps(e), flush=True)
