#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: Wifi80211 link (64-QAM, K=7 convolutional code, soft Viterbi) Monte-Carlo
BER sweep Eb/N0 = 0..10 dB, GPU-resident chain (commpy_amd.devicelink.DeviceWifiLink).

    python benchmarks/bench_link.py [--mcs 5] [--bits 1e8] [--generators octal|decimal]

Prints one JSON line: end-to-end simulated info-bits/s (all stages: bits, encode, puncture, modulate, AWGN,
soft demod, depuncture, Viterbi, error count) and the BER curve.  SNR_dB = Eb/N0 + 10 log10(Rc * Mc) as in
the reference's docstrings (wifi80211.py:139-141); quirk B7 makes the effective Eb/N0 3 dB lower.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mcs", type=int, default=5)
    ap.add_argument("--bits", type=float, default=1e8)
    ap.add_argument("--generators", default="octal")
    ap.add_argument("--send-chunk", type=int, default=1200)
    a = ap.parse_args()
    from commpy_amd.devicelink import DeviceWifiLink
    gm = [[0o133, 0o171]] if a.generators == "octal" else None
    link = DeviceWifiLink(a.mcs, a.send_chunk, generator_matrix=gm, seed=2026)
    ebn0 = np.arange(0.0, 10.5, 1.0)
    snrs = ebn0 + 10 * math.log10(link.rate * link.modem.num_bits_symbol)
    per_point = int(a.bits / len(snrs))
    link.ber_sweep_batched(snrs, per_point)              # warm-up: same buffers as the timed sweep (allocations, code objects)
    t0 = time.perf_counter()
    bers = link.ber_sweep_batched(snrs, per_point)       # one Viterbi call over all SNR points (fused large-batch kernel)
    dt = time.perf_counter() - t0
    total = len(snrs) * math.ceil(per_point / link.nbits) * link.nbits
    print(json.dumps({
        "benchmark": "config 5: Wifi80211 MCS%d (%d-QAM, rate %d/%d), %s generators, Eb/N0 0..10 dB" % (
            a.mcs, link.modem.m, link.coding[0], link.coding[1], a.generators),
        "value": total / dt, "unit": "simulated info-bits/s (end to end, 1 GPU)", "seconds": dt,
        "info_bits": total, "frame_bits": link.nbits,
        "ebn0_db": ebn0.tolist(), "snr_db": [round(float(s), 3) for s in snrs], "ber": bers.tolist(),
        "reference_cpu": "Wifi80211(mcs=5).link_performance: ~1.0e3 info-bits/s on one core (BASELINE.md)"}), flush=True)


if __name__ == "__main__":
    main()
