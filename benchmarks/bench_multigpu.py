#!/usr/bin/env python3
"""BASELINE configs 4 and 5 sharded over the GPUs of one node by ONE process (commpy_amd.parallel.DeviceGroup:
one host thread and one stream per device, RCCL collectives through the engine's C-ABI).

    python benchmarks/bench_multigpu.py [--gpus G] [--which config4,config5] [--blocks 262144] [--bits 1e8]

config 4   B blocks of the (1944,1296) code -> B/G per GPU (BASELINE: 262144 -> 32768 on each of 8 GPUs).  Every GPU
           generates its share on the device (random messages -> systematic encoder -> 64-QAM -> AWGN -> soft demod ->
           sign flip), decodes it (min-sum and sum-product, <= 50 iterations) and the int8 dec_word of all shards is
           all-gathered over xGMI (one RCCL all-gather per decode, in place) -- the timed region is decode + gather.
config 5   Wifi80211 MCS 5 sweep, Eb/N0 = 0..10 dB, `bits` information bits per sweep split over the GPUs; error and
           bit counters are summed with one RCCL all-reduce of int64 counters (links.py:252-260).
One JSON line per measurement; `n_gpus` says what it was measured on (the build box has one GPU: G = 1 exercises the
same code path, collectives included, with a communicator of one rank).  No hardware scaling curve exists yet.
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from commpy_amd import _lib  # noqa: E402


def config4(grp, B_total, ebn0=9.0):
    from commpy_amd.channelcoding.ldpc import _device_code, get_ldpc_code_params
    from commpy_amd.devicelink import DeviceBuf, LdpcEncoder
    from commpy_amd.modulation import QAMModem
    from commpy_amd.parallel import shard_counts
    lib = grp.lib
    p = get_ldpc_code_params(os.path.join(ROOT, "commpy_amd/channelcoding/designs/ldpc/ieee80211n/1944.1296.txt"), True)
    md = QAMModem(64)
    n, nsym, k = 1944, 324, 1296
    counts = shard_counts(B_total, grp.G)
    rows = max(counts)
    N0 = 42.0 / ((2.0 / 3) * 6 * 10 ** (ebn0 / 10.0))
    sc = float(np.sqrt(N0 / 2))
    grp.prepare(lambda: (_device_code(p), md._device_handle()))

    def setup(i, dev, st):
        B = counts[i]
        enc = LdpcEncoder(p, "gf2")
        bufs = dict(msg=DeviceBuf(B * k), bits=DeviceBuf(B * n), sym=DeviceBuf(B * nsym * 16), y=DeviceBuf(B * nsym * 16),
                    llr=DeviceBuf(B * n * 8), neg=DeviceBuf(B * n * 8), out=DeviceBuf(B * n * 8), it=DeviceBuf(B * 4),
                    dec=DeviceBuf(grp.G * rows * n), enc=enc)
        ck = _lib.check
        ck(lib.cpx_random_bits_dev(bufs["msg"].ptr, B * k, 30 + i, 0, st))
        ck(lib.cpx_ldpc_encode_batch_dev(enc.h, bufs["msg"].ptr, B, bufs["bits"].ptr, st))
        ck(lib.cpx_modulate_dev(md._device_handle(), bufs["bits"].ptr, B * nsym, bufs["sym"].ptr, st))
        ck(lib.cpx_awgn_dev(bufs["sym"].ptr, B * nsym, sc, sc, 31 + i, 1, bufs["y"].ptr, st))
        # demodulate returns log P1/P0, the decoder takes log P0/P1 (test_ldpc.py:53-54): the flip rides in the demodulator
        ck(lib.cpx_demod_soft_scaled_dev(md._device_handle(), bufs["y"].ptr, B * nsym, float(N0), -1.0, bufs["llr"].ptr, st))
        ck(lib.cpx_stream_sync(st))
        return bufs

    bufs = grp.each(setup)
    for alg, name in ((1, "MSA"), (0, "SPA")):
        def step():
            def launch(i, dev, st):
                B = counts[i]
                b = bufs[i]
                mine = ctypes.c_void_p(b["dec"].ptr.value + i * rows * n)
                _lib.check(lib.cpx_memcpy_d2d_async(b["neg"].ptr, b["llr"].ptr, B * n * 8, st))    # the decoder clips in place
                _lib.check(lib.cpx_ldpc_bp_decode_batch_bm_dev(_device_code(p), b["neg"].ptr, B, alg, 50, mine, b["out"].ptr,
                                                            b["it"].ptr, st))
                return mine
            mine = grp.each(launch)
            grp.allgather_dev(mine, [b["dec"].ptr for b in bufs], rows * n)      # dec_word of every shard on every GPU
            grp.sync()
        step()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            step()
        dt = (time.perf_counter() - t0) / reps
        # check on GPU 0: its own shard and, through the gathered buffer, the last GPU's shard against what was sent
        def check(i, dev, st):
            sent = bufs[i]["bits"].to_array((counts[i], n), np.int8)
            its = bufs[i]["it"].to_array((counts[i],), np.int32)
            return sent, its
        sent_its = grp.each(check)
        full = grp.each(lambda i, dev, st: bufs[i]["dec"].to_array((grp.G, rows * n), np.int8) if i == 0 else None)[0]
        fer = []
        for g in range(grp.G):
            dec = full[g, :n * counts[g]].reshape(counts[g], n)       # block-major shards: one block per row
            fer.append(float(np.mean((dec != sent_its[g][0]).any(axis=1))))
        its = np.concatenate([s[1] for s in sent_its])
        print(json.dumps({"benchmark": "config 4 sharded: (1944,1296) LDPC %s, 64-QAM soft demod at Eb/N0 = %.0f dB, <= 50 its, "
                                       "B = %d blocks over %d GPU(s), dec_word all-gathered (RCCL)" % (name, ebn0, B_total, grp.G),
                          "value": B_total * 1296 / dt, "unit": "info-bits/s (decode + all-gather, whole job)", "ms": dt * 1e3,
                          "n_gpus": grp.G, "blocks_per_gpu": counts, "mean_iterations": float(its.mean()),
                          "frame_error_rate_per_shard_seen_from_gpu0": fer,
                          "gathered_bytes_per_gpu": int(grp.G * rows * n)}), flush=True)
    for b in bufs:
        for v in b.values():
            if hasattr(v, "free"):
                v.free()


def config5(grp, bits):
    ebn0 = np.arange(0.0, 10.5, 1.0)
    snrs = ebn0 + 10 * math.log10((2.0 / 3) * 6)
    per_point = int(bits / len(snrs))
    grp.wifi_ber_sweep(5, snrs, per_point, send_chunk=1200, generator_matrix=[[0o133, 0o171]], seed=2026)   # warm-up
    t0 = time.perf_counter()
    ber, errs, nbits = grp.wifi_ber_sweep(5, snrs, per_point, send_chunk=1200, generator_matrix=[[0o133, 0o171]], seed=2026)
    dt = time.perf_counter() - t0
    print(json.dumps({"benchmark": "config 5 sharded: Wifi80211 MCS5 (64-QAM, rate 2/3), octal generators, Eb/N0 0..10 dB, "
                                   "counters all-reduced (RCCL) over %d GPU(s)" % grp.G,
                      "value": float(nbits.sum()) / dt, "unit": "simulated info-bits/s (end to end, whole job)", "seconds": dt,
                      "n_gpus": grp.G, "info_bits": int(nbits.sum()), "ebn0_db": ebn0.tolist(), "ber": ber.tolist(),
                      "note": "includes creating the per-GPU link objects and buffers of the sweep (one sweep = one call)"}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=0, help="0 = all visible")
    ap.add_argument("--which", default="config4,config5")
    ap.add_argument("--blocks", type=int, default=0, help="config 4 blocks in total (default 32768 per GPU)")
    ap.add_argument("--bits", type=float, default=1e8)
    a = ap.parse_args()
    from commpy_amd.parallel import DeviceGroup
    _lib.load()
    _lib.require_device()
    G = a.gpus or _lib.device_count()
    grp = DeviceGroup(list(range(G)))
    if "config4" in a.which:
        config4(grp, a.blocks or 32768 * G)
    if "config5" in a.which:
        config5(grp, a.bits)
    grp.close()


if __name__ == "__main__":
    main()
