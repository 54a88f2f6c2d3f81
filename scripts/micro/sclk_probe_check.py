#!/usr/bin/env python3
"""Does the shader-clock probe (cpx_sclk_probe_*: one sleeping wavefront on its own stream) disturb what it measures?  The config-2
Viterbi batch, K launches per repetition, alternately with and without the probe alongside, after a real warm-up; prints per-launch
medians and the probe's clock.  (round 5; bench.py runs its probe in an extra, untimed repetition of its K steps.)"""
import ctypes
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from commpy_amd import _lib  # noqa: E402
from benchmarks.other_configs import Dev, Timers, warm  # noqa: E402
from commpy_amd.channelcoding import Trellis  # noqa: E402

lib = _lib.load()
tr = Trellis(np.array([6]), np.array([[0o133, 0o171]]))
B, K = 65536, 20
rs = np.random.RandomState(1)
dev = Dev(lib)
llr = rs.standard_normal((B, 2060)).astype(np.float64) * 2.0
d_in, d_out = dev.put(llr), dev.empty(B * 1030)
h = tr._device_handle()


def step():
    _lib.check(lib.cpx_viterbi_decode_batch_dev(h, d_in, B, 2060, 1030, 1035, 30, 1, d_out, None))


warm(lib, step, 5, 0.1)
for rep in range(4):
    for with_probe in (0, 1):
        probe = ctypes.c_void_p()
        if with_probe:
            _lib.check(lib.cpx_sclk_probe_start(ctypes.byref(probe), 0.9 * K * 1.55))
        tm = Timers(lib, K)
        for i in range(K):
            tm.start(i); step(); tm.stop(i)
        _lib.check(lib.cpx_stream_sync(None))
        ms = tm.read()
        txt = ""
        if with_probe:
            mhz, iv = ctypes.c_double(), ctypes.c_double()
            _lib.check(lib.cpx_sclk_probe_read(probe, ctypes.byref(mhz), ctypes.byref(iv)))
            txt = "sclk %.0f MHz over %.1f ms -> %.3f M cycles per launch" % (mhz.value, iv.value, mhz.value * 1e-3 * np.median(ms))
        print("rep %d probe %d: median %.4f ms  mean %.4f  min %.4f   %s" % (rep, with_probe, np.median(ms), ms.mean(), ms.min(), txt), flush=True)
dev.free()
