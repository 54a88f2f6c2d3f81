import ctypes, sys, numpy as np
sys.path.insert(0, '.')
from commpy_amd import _lib
from commpy_amd.channelcoding import Trellis
from commpy_amd.devicelink import DeviceBuf
lib = _lib.load()
tr = Trellis(np.array([6]), np.array([[0o133, 0o171]]))
B, n, L, T = 65536, 2060, 1030, 1035
x = np.random.RandomState(0).randn(B, n) * 4
d_in, d_out = DeviceBuf.from_array(x), DeviceBuf(B * L)
tm = ctypes.c_void_p(); lib.cpx_timer_create(ctypes.byref(tm))
for tb in (30, 15, 29, 2, 40):
    best = 1e9
    for i in range(4):
        lib.cpx_timer_start(tm, None)
        _lib.check(lib.cpx_viterbi_decode_batch_dev(tr._device_handle(), d_in.ptr, B, n, L, T, tb, 1, d_out.ptr, None))
        lib.cpx_timer_stop(tm, None)
        v = ctypes.c_float(); lib.cpx_timer_elapsed_ms(tm, ctypes.byref(v))
        if i: best = min(best, v.value)
    print("tb=%d: %.3f ms  %s" % (tb, best, _lib.last_kernel()), flush=True)
