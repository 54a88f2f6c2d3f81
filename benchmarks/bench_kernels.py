#!/usr/bin/env python3
"""Secondary benchmarks: the other kernels of the hot path on their BASELINE.json configurations
(device-resident inputs, HIP-event timing on the launch stream).  One JSON line per kernel.

    python benchmarks/bench_kernels.py [--which demod,ldpc,turbo,map,viterbi_small,viterbi_variants,viterbi_k9,config4,encoders] [--scale 1.0]

  demod   64-QAM soft LLR, 324 symbols x 32768 codewords (config 4's per-GPU share)   -> HBM roofline
  ldpc    (1944,1296) BP, 50 iterations max, B = 32768 (config 4 per-GPU share), SPA and MSA
  turbo   rate-1/3, N = 1024, 6 iterations, random interleaver, B = 16384 (config 3)
  map     one MAP pass of the same
  viterbi_small  config 1 (K=3, 64-bit, hard) scaled to B = 1M codewords
  encoders  device turbo_encode (config 3) and systematic LDPC encode (config 4)
bench.py (repo root) stays the headline Viterbi benchmark.
"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from commpy_amd import _lib  # noqa: E402

HBM_PEAK = 8000.0


from benchmarks.other_configs import Dev, warm  # noqa: E402  (the helpers of bench.py's `other_configs` lines)


def timeit(lib, fn, steps=5, warmup=1):
    """Mean / min milliseconds of `steps` calls, HIP events on the launch stream, after a warm-up that also covers the clock ramp of a
    fresh process (round 5; round 4 timed 3 - 5 calls after ONE warm-up call and read up to 9 % high)."""
    tm = ctypes.c_void_p()
    _lib.check(lib.cpx_timer_create(ctypes.byref(tm)))
    warm(lib, fn, warmup)
    ms = []
    for _ in range(steps):
        _lib.check(lib.cpx_timer_start(tm, None))
        fn()
        _lib.check(lib.cpx_timer_stop(tm, None))
        v = ctypes.c_float()
        _lib.check(lib.cpx_timer_elapsed_ms(tm, ctypes.byref(v)))
        ms.append(v.value)
    lib.cpx_timer_destroy(tm)
    return float(np.mean(ms)), float(np.min(ms))


def emit(name, workload, units, unit_name, ms, alg_bytes, bound, extra=None, dtype="f64"):
    ach = alg_bytes / (ms * 1e-3) / 1e9
    d = {"kernel": name, "workload": workload, "value": units / (ms * 1e-3), "unit": unit_name + "/s", "ms": ms,
         "dtype": dtype, "roofline": {"bound": bound, "achieved": ach, "peak": HBM_PEAK, "unit": "GB/s",
                                      "frac": ach / HBM_PEAK, "algorithmic_bytes_per_launch": alg_bytes}}
    if extra:
        d.update(extra)
    print(json.dumps(d), flush=True)


def bench_demod(lib, scale):
    from commpy_amd.modulation import QAMModem
    for m, ncw, per_cw in ((64, int(32768 * scale), 324), (4, int(65536 * scale), 1030)):
        md = QAMModem(m)
        ns = ncw * per_cw
        rs = np.random.RandomState(31)
        N0 = md.Es / ((2.0 / 3) * md.num_bits_symbol * 10 ** 0.8) if m == 64 else md.Es / (0.5 * 2 * 10 ** 0.3)
        y = md.constellation[rs.randint(0, m, ns)] + np.sqrt(N0 / 2) * (rs.randn(ns) + 1j * rs.randn(ns))
        dev = Dev(lib)
        d_y = dev.put(y)
        d_l = dev.empty(ns * md.num_bits_symbol * 8)
        d_b = dev.empty(ns * md.num_bits_symbol)
        h = md._device_handle()
        ms, _ = timeit(lib, lambda: _lib.check(lib.cpx_demod_soft_dev(h, d_y, ns, float(N0), d_l, None)))
        emit(_lib.last_kernel(), "%d-QAM soft LLR, %d symbols" % (m, ns), ns, "symbols", ms,
             ns * (16 + 8 * md.num_bits_symbol), "valu")
        ms, _ = timeit(lib, lambda: _lib.check(lib.cpx_demod_hard_dev(h, d_y, ns, d_b, None)))
        emit(_lib.last_kernel(), "%d-QAM hard decision, %d symbols" % (m, ns), ns, "symbols", ms,
             ns * (16 + md.num_bits_symbol), "hbm")
        dev.free()
    # generic constellations (round 6: demod_soft_gen_kernel): two inputs used in turn so that every byte comes from HBM (2 x 170 MB
    # of symbols > the 256 MiB Infinity Cache), and the literal kernel ("libm") beside it
    from commpy_amd.modulation import PSKModem
    for m, snr_db in ((8, 10.0), (16, 16.0)):
        md = PSKModem(m)
        nb = md.num_bits_symbol
        ns = int(10616832 * scale)
        rs = np.random.RandomState(32)
        N0 = md.Es / 10 ** (snr_db / 10.0)
        dev = Dev(lib)
        d_ys = [dev.put(md.constellation[rs.randint(0, m, ns)] + np.sqrt(N0 / 2) * (rs.randn(ns) + 1j * rs.randn(ns))) for _ in range(2)]
        d_ls = [dev.empty(ns * nb * 8) for _ in range(2)]
        h = md._device_handle()
        for mode in (None, "libm"):
            _lib.demod_set_path(mode)
            try:
                cnt = [0]

                def step():
                    cnt[0] += 1
                    _lib.check(lib.cpx_demod_soft_dev(h, d_ys[cnt[0] % 2], ns, float(N0), d_ls[cnt[0] % 2], None))
                ms, _ = timeit(lib, step, steps=6, warmup=2)
                emit(_lib.last_kernel(), "%d-PSK soft LLR, %d symbols, HBM-resident (2 inputs in turn)" % (m, ns), ns, "symbols", ms,
                     ns * (16 + 8 * nb), "hbm + valu")
            finally:
                _lib.demod_set_path(None)
        dev.free()


def bench_ldpc(lib, scale):
    from commpy_amd.channelcoding.ldpc import _device_code, get_ldpc_code_params
    p = get_ldpc_code_params(os.path.join(ROOT, "commpy_amd/channelcoding/designs/ldpc/ieee80211n/1944.1296.txt"), True)
    n, E = 1944, 7128
    B = int(32768 * scale)
    rs = np.random.RandomState(31)
    for ebn0 in (2.2, 3.0):
        sigma = 1 / np.sqrt(10 ** (ebn0 / 10.0) * (2.0 / 3) * 2)
        llr = (2.0 * (1.0 + sigma * rs.randn(B, n)) / sigma ** 2)
        dev = Dev(lib)
        d_llr0 = dev.put(llr)
        d_llr = dev.empty(llr.nbytes)
        d_dec, d_out, d_it = dev.empty(B * n), dev.empty(B * n * 8), dev.empty(B * 4)
        code = _device_code(p)
        for alg, name in ((1, "MSA"), (0, "SPA")):
            def run():
                _lib.check(lib.cpx_memcpy_h2d(d_llr, _lib.ptr(llr), llr.nbytes))   # decode clips in place: restore input
                _lib.check(lib.cpx_ldpc_bp_decode_batch_bm_dev(code, d_llr, B, alg, 50, d_dec, d_out, d_it, None))

            tm = ctypes.c_void_p()
            lib.cpx_timer_create(ctypes.byref(tm))
            run()
            _lib.check(lib.cpx_stream_sync(None))
            _lib.check(lib.cpx_memcpy_h2d(d_llr, _lib.ptr(llr), llr.nbytes))
            lib.cpx_timer_start(tm, None)
            _lib.check(lib.cpx_ldpc_bp_decode_batch_bm_dev(code, d_llr, B, alg, 50, d_dec, d_out, d_it, None))
            lib.cpx_timer_stop(tm, None)
            v = ctypes.c_float()
            lib.cpx_timer_elapsed_ms(tm, ctypes.byref(v))
            its = dev.get(d_it, (B,), np.int32)
            dec = dev.get(d_dec, (B, n), np.int8).T          # block-major device layout = the reference's F-ordered result
            # Bytes (review of round 2: the per-iteration re-read of the channel LLRs is an L2 hit, not HBM traffic, and must not be
            # counted).  The fraction is quoted on SURVEY 8d's RESIDENT model -- what a decoder whose messages never leave the
            # chip has to move: llr in (8 n), out_llrs (8 n) and dec_word (n) out = 17 n B per block, whatever the number of
            # iterations.  With block-major outputs (this call) that is also what the engine asks of HBM: retired blocks go straight
            # to the caller's arrays, there is no staging buffer and no transposition pass.  SURVEY 8d's streaming figure of the
            # reference formulation rides along for scale.  The kernel is bound by LDS bandwidth (min-sum) / VALU issue
            # (sum-product), PMC in profiles/: the HBM fraction is reported because the metric asks for it.
            stream_bytes = int(its.sum()) * (4 * E + 2 * n) * 8 + B * n * 17
            res_bytes = B * n * 17
            eng_bytes = B * n * 17
            kname = _lib.last_kernel()
            emit("ldpc_bp_%s" % name, "(1944,1296) Eb/N0=%.1f dB, <=50 its, B=%d, mean executed its %.2f" % (
                ebn0, B, its.mean()), B * 1296, "info-bits", v.value, res_bytes, "valu+lds" if "resident" in kname else "hbm",
                 {"kernel_name": kname,
                  "frame_error_rate": float(np.mean(dec.any(axis=0))), "mean_iterations": float(its.mean()),
                  "max_iterations": int(its.max()),
                  "block_iterations_per_s": float(its.sum()) / (v.value * 1e-3),
                  "bytes_model": "SURVEY 8d resident model: 17 n B per block (llr in, out_llrs + dec_word out); block-major "
                                 "outputs, no staging",
                  "survey_8d_streaming_formulation_bytes_per_launch": stream_bytes})
        dev.free()


def bench_config4(lib, scale):
    """BASELINE config 4 on one GPU's share: random messages -> systematic (1944,1296) encoder -> 64-QAM -> AWGN ->
    soft demod -> sign flip (quirk B6) -> LDPC BP (<= 50 iterations), every stage on the device."""
    from commpy_amd.devicelink import LdpcEncoder
    from commpy_amd.channelcoding.ldpc import _device_code, get_ldpc_code_params
    from commpy_amd.modulation import QAMModem
    p = get_ldpc_code_params(os.path.join(ROOT, "commpy_amd/channelcoding/designs/ldpc/ieee80211n/1944.1296.txt"), True)
    md = QAMModem(64)
    n, nsym = 1944, 324
    B = int(32768 * scale)
    dev = Dev(lib)
    enc = LdpcEncoder(p, "gf2")
    d_msg, d_bits = dev.empty(B * enc.k), dev.empty(B * n)
    d_sym, d_y = dev.empty(B * nsym * 16), dev.empty(B * nsym * 16)
    d_llr, d_neg = dev.empty(B * n * 8), dev.empty(B * n * 8)
    d_dec, d_out, d_it = dev.empty(B * n), dev.empty(B * n * 8), dev.empty(B * 4)
    code, h_md = _device_code(p), md._device_handle()
    for ebn0 in (8.0, 9.0, 10.0):
        N0 = 42.0 / ((2.0 / 3) * 6 * 10 ** (ebn0 / 10.0))
        sc = float(np.sqrt(N0 / 2))
        for alg, name in ((1, "MSA"), (0, "SPA")):
            def run():
                _lib.check(lib.cpx_random_bits_dev(d_msg, B * enc.k, 30, 0, None))
                enc.encode_dev(d_msg, B, d_bits)
                _lib.check(lib.cpx_modulate_dev(h_md, d_bits, B * nsym, d_sym, None))
                _lib.check(lib.cpx_awgn_dev(d_sym, B * nsym, sc, sc, 31, 1, d_y, None))
                # the sign flip of quirk B6 (demodulate returns log P1/P0, the decoder takes log P0/P1) rides in the demodulator
                _lib.check(lib.cpx_demod_soft_scaled_dev(h_md, d_y, B * nsym, float(N0), -1.0, d_neg, None))
                _lib.check(lib.cpx_ldpc_bp_decode_batch_bm_dev(code, d_neg, B, alg, 50, d_dec, d_out, d_it, None))
            ms, _ = timeit(lib, run, steps=3, warmup=1)
            its = dev.get(d_it, (B,), np.int32)
            dec = dev.get(d_dec, (B, n), np.int8).T
            sent = dev.get(d_bits, (B, n), np.int8)
            # bytes the chain moves through HBM: symbols written / read / written with noise (16 B each), 6 LLRs per symbol (48 B)
            # written by the demodulator, and the decoder's resident-model bytes (17 n: llr in, out_llrs + dec_word out)
            alg_bytes = B * nsym * (16 * 3 + 48) + B * n * 17
            kname = _lib.last_kernel()
            emit("config4_pipeline_%s" % name, "encode + 64-QAM + AWGN + demod + LDPC (1944,1296) %s, Eb/N0=%.0f dB, B=%d, "
                 "mean its %.2f" % (name, ebn0, B, its.mean()), B * 1296, "info-bits", ms, alg_bytes,
                 "valu+lds" if "resident" in kname else "hbm",
                 {"decoder_kernel": kname,
                  "frame_error_rate": float(np.mean((dec.T != sent).any(axis=1))),
                  "bit_error_rate": float(np.mean(dec.T[:, :1296] != sent[:, :1296])),
                  "mean_iterations": float(its.mean())})
    dev.free()


def bench_turbo(lib, scale, which, states=4):
    from benchmarks.other_configs import turbo_workload          # the same workload bench.py's `other_configs` line times
    N, B = 1024, int(16384 * scale)
    tr, il, msgs, s, p1, p2, nv = turbo_workload(B, N, states)
    dev = Dev(lib)
    d_s, d_p1, d_p2 = dev.put(s), dev.put(p1), dev.put(p2)
    d_perm = dev.put(np.asarray(il.p_array, dtype=np.int32))
    d_bits = dev.empty(B * N)
    d_L = dev.empty(B * N * 8)
    d_zero = dev.put(np.zeros((B, N)))
    h = tr._device_handle()
    if "turbo" in which:
        ms, _ = timeit(lib, lambda: _lib.check(lib.cpx_turbo_decode_batch_dev(h, d_s, d_p1, d_p2, None, d_perm, B, N, nv, 6,
                                                                             d_bits, None)), steps=3)
        bits = dev.get(d_bits, (B, N), np.uint8)
        emit("turbo_decode (turbo_pass_kernel x 12 + init + final)", "rate-1/3 %d-state RSC, N=1024, 6 iterations, Eb/N0=1.5 dB, B=%d" % (states, B), B * N,
             "info-bits", ms, B * 25600, "latency/valu", {"ber": float(np.mean(bits != msgs))})
    if "map" in which:
        ms, _ = timeit(lib, lambda: _lib.check(lib.cpx_map_decode_batch_dev(h, d_s, d_p1, d_zero, B, N, nv, 1, d_L, d_bits,
                                                                           None)), steps=3)
        emit("map_decode_kernel", "one MAP pass, %d-state RSC, N=1024, B=%d" % (states, B), B * N, "info-bits", ms, B * 33792,
             "latency/valu")
    dev.free()


def bench_viterbi_variants(lib, scale):
    """The fused codeword-per-lane kernel away from the headline instantiation, on the config-2 geometry (65536 x 1024-bit, soft):
    traceback depth 40 (64-slot ring), a pair without a compiled-in instantiation (table-driven), and the state-per-lane kernels."""
    from commpy_amd.channelcoding import Trellis, conv_encode_batch
    B = int(65536 * scale)
    rs = np.random.RandomState(4)
    for gm, tb, path, what in (([0o133, 0o171], 40, None, "tb_depth = 40"), ([0o135, 0o147], 30, None, "(135,147), no compiled-in instantiation"),
                               ([0o135, 0o147], 30, "jit", "(135,147), its own code object (Trellis.specialize, commpy_amd/jit.py)"),
                               ([0o133, 0o171], 30, "wave", "state-per-lane kernels")):
        tr = Trellis(np.array([6]), np.array([gm]))
        if path == "jit":
            path = None
            if not tr.specialize():
                print("# (135,147): no code object (%s)" % __import__("commpy_amd.jit", fromlist=["x"]).viterbi_code_object.last_error, flush=True)
                continue
        coded = conv_encode_batch(rs.randint(0, 2, (B, 1024)).astype(np.uint8), tr).astype(np.float64)
        llr = np.ascontiguousarray(4.0 * coded - 2 + rs.standard_normal(coded.shape).astype(np.float32) * 1.4, dtype=np.float64)
        dev = Dev(lib)
        d_in, d_out = dev.put(llr), dev.empty(B * 1030)
        h = tr._device_handle()
        _lib.viterbi_set_path(path)
        try:
            ms, _ = timeit(lib, lambda: _lib.check(lib.cpx_viterbi_decode_batch_dev(h, d_in, B, 2060, 1030, 1030, tb, 1, d_out, None)), steps=10)
            name = _lib.last_kernel()
        finally:
            _lib.viterbi_set_path(None)
        emit(name, "config-2 geometry, %s, B=%d" % (what, B), B * 1024, "info-bits", ms, B * 17510, "valu")
        dev.free()


def bench_viterbi_k9(lib, scale):
    """K = 9 (561,753), 256 states (round 4): four states per lane on the wide kernel, and the general kernel on a smaller batch."""
    from commpy_amd.channelcoding import Trellis, conv_encode_batch
    tr = Trellis(np.array([8]), np.array([[0o561, 0o753]]))
    rs = np.random.RandomState(9)
    for path, B in ((None, int(16384 * scale)), ("general", int(2048 * scale))):
        coded = conv_encode_batch(rs.randint(0, 2, (B, 1024)).astype(np.uint8), tr).astype(np.float64)      # 2064 values, 1032 bits
        llr = np.ascontiguousarray(4.0 * coded - 2 + rs.standard_normal(coded.shape).astype(np.float32) * 1.4, dtype=np.float64)
        L, T = coded.shape[1] // 2, coded.shape[1] // 2 + 8 - 1
        dev = Dev(lib)
        d_in, d_out = dev.put(llr), dev.empty(B * L)
        h = tr._device_handle()
        _lib.viterbi_set_path(path)
        try:
            ms, _ = timeit(lib, lambda: _lib.check(lib.cpx_viterbi_decode_batch_dev(h, d_in, B, coded.shape[1], L, T, 40, 1, d_out, None)), steps=3)
            name = _lib.last_kernel()
        finally:
            _lib.viterbi_set_path(None)
        emit(name, "K=9 (561,753) r=1/2, 1024-bit blocks, soft, tb_depth 40, B=%d" % B, B * 1024, "info-bits", ms, B * (coded.shape[1] * 8 + L), "valu")
        dev.free()


def bench_viterbi_small(lib, scale):
    """BASELINE config 1, end to end on the device (round 5): Philox messages -> conv_encode('term') -> cpx_bsc_dev(p = 0.05) -> hard
    Viterbi (commpy_amd.devicelink.DeviceBscLink); the decode alone is what is timed."""
    from commpy_amd.channelcoding import Trellis
    from commpy_amd.devicelink import DeviceBscLink
    tr = Trellis(np.array([2]), np.array([[5, 7]]))
    B = int((1 << 20) * scale)
    link = DeviceBscLink(tr, 64, tb_depth=10, seed=1)
    errs = link.run_batch(0.05, B)                                 # generate + decode + count once: the BER of what is timed below
    ms, _ = timeit(lib, lambda: link.decode(B))
    emit(_lib.last_kernel(), "config 1: K=3 [[5,7]] 64-bit blocks, hard/BSC(0.05), device-generated, B=%d" % B, B * 64,
         "info-bits", ms, B * (link.ncoded * 8 + link.L), "valu", {"ber": float(errs.sum()) / (B * 64.0)})
    ms, _ = timeit(lib, lambda: link.generate(0.05, B))
    emit("random_bits + conv_encode_ff + binary_channel_kernel<false>", "config 1 generator: messages, encoder, BSC on the device, B=%d" % B,
         B * 64, "info-bits", ms, B * (64 + 2 * link.ncoded + 8 * link.ncoded), "hbm", dtype="u8")


def bench_encoders(lib, scale):
    """the device encoders of configs 3 and 4 (byte streams: HBM roofline)."""
    import warnings
    from commpy_amd.channelcoding import RandInterlv, Trellis
    from commpy_amd.channelcoding.ldpc import get_ldpc_code_params
    from commpy_amd.devicelink import LdpcEncoder
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tr = Trellis(np.array([2]), np.array([[1, 7]]), 5, "rsc")
    N, B = 1024, int(16384 * scale)
    il = RandInterlv(N, 1234)
    dev = Dev(lib)
    d_msg = dev.put(np.random.RandomState(20).randint(0, 2, (B, N)).astype(np.uint8))
    d_perm = dev.put(np.asarray(il.p_array, dtype=np.int32))
    d_s, d_p1, d_p2 = dev.empty(B * N), dev.empty(B * N), dev.empty(B * N)
    h = tr._device_handle()
    for mode, name in ((2, "turbo_encode_wave_kernel<4>"), (1, "turbo_encode_seq_kernel")):
        ms, _ = timeit(lib, lambda: _lib.check(lib.cpx_turbo_encode_batch_dev(h, h, d_msg, B, N, d_perm, d_s, d_p1, d_p2, N,
                                                                             mode, None)))
        emit(name, "turbo_encode, 4-state RSC, N=1024, B=%d" % B, B * N, "info-bits", ms, B * 4 * N, "hbm", dtype="u8")
    dev.free()
    design = os.path.join(ROOT, "commpy_amd", "channelcoding", "designs", "ldpc", "ieee80211n", "1944.1296.txt")
    enc = LdpcEncoder(get_ldpc_code_params(design), "gf2")
    B = int(32768 * scale)
    dev = Dev(lib)
    d_msg = dev.put(np.random.RandomState(31).randint(0, 2, (B, enc.k)).astype(np.uint8))
    d_code = dev.empty(B * enc.n)
    ms, _ = timeit(lib, lambda: enc.encode_dev(d_msg, B, d_code))
    emit("ldpc_pack_kernel+ldpc_parity_kernel", "systematic LDPC encode (1944,1296), B=%d" % B, B * enc.k, "info-bits", ms,
         B * (enc.k + enc.n), "hbm", dtype="u8")
    dev.free()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default="demod,ldpc,config4,turbo,map,turbo8,viterbi_small,viterbi_variants,viterbi_k9,encoders")
    ap.add_argument("--scale", type=float, default=1.0)
    a = ap.parse_args()
    lib = _lib.load()
    _lib.require_device()
    which = a.which.split(",")
    if "demod" in which:
        bench_demod(lib, a.scale)
    if "viterbi_variants" in which:
        bench_viterbi_variants(lib, a.scale)
    if "viterbi_k9" in which:
        bench_viterbi_k9(lib, a.scale)
    if "viterbi_small" in which:
        bench_viterbi_small(lib, a.scale)
    if "turbo" in which or "map" in which:
        bench_turbo(lib, a.scale, which)
    if "turbo8" in which:
        bench_turbo(lib, a.scale, ["turbo", "map"], states=8)
    if "ldpc" in which:
        bench_ldpc(lib, a.scale)
    if "config4" in which:
        bench_config4(lib, a.scale)
    if "encoders" in which:
        bench_encoders(lib, a.scale)


if __name__ == "__main__":
    main()
