import ctypes, sys, os
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/benchmarks")
from commpy_amd import _lib
from bench_kernels import Dev
from commpy_amd.channelcoding import Trellis
lib = _lib.load()
tr = Trellis(np.array([6]), np.array([[0o133, 0o171]]))
h = tr._device_handle()
rs = np.random.RandomState(0)
LEN, L = 2060, 1030
for B in (30000, 34000, 38000, 45000, 65536 + 34000, 65536 + 38000):
    x = rs.randn(B, LEN) * 3
    dev = Dev(lib)
    d_in, d_out = dev.put(x), dev.empty(B * L)
    tm = ctypes.c_void_p(); lib.cpx_timer_create(ctypes.byref(tm))
    res = []
    for path in (None, "wave", "cw"):
        _lib.viterbi_set_path(path)
        best = 1e9
        for rep in range(3):
            lib.cpx_timer_start(tm, None)
            _lib.check(lib.cpx_viterbi_decode_batch_dev(h, d_in, B, LEN, L, L, 30, 1, d_out, None))
            lib.cpx_timer_stop(tm, None)
            v = ctypes.c_float(); lib.cpx_timer_elapsed_ms(tm, ctypes.byref(v)); best = min(best, v.value)
        res.append("%s %.3f ms [%s]" % (path or "auto", best, _lib.viterbi_last_path()))
    _lib.viterbi_set_path(None)
    print(B, " | ".join(res), flush=True)
    dev.free()
