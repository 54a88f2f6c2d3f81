"""Times cpx_viterbi_decode_batch_dev (K=7, soft, B=65536 x 1024 bits) on different input distributions and paths."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from commpy_amd import _lib
from commpy_amd.channelcoding import Trellis
from commpy_amd.devicelink import DeviceBuf
lib = _lib.load()
tr = Trellis(np.array([6]), np.array([[0o133, 0o171]]))
B, n, L, T = 65536, 2060, 1030, 1035
rs = np.random.RandomState(0)
def timeit(d_in, d_out, path, reps=5):
    _lib.viterbi_set_path(path)
    tm = ctypes.c_void_p(); lib.cpx_timer_create(ctypes.byref(tm))
    best = 1e9
    for i in range(reps + 1):
        lib.cpx_timer_start(tm, None)
        _lib.check(lib.cpx_viterbi_decode_batch_dev(tr._device_handle(), d_in.ptr, B, n, L, T, 30, 1, d_out.ptr, None))
        lib.cpx_timer_stop(tm, None)
        v = ctypes.c_float(); lib.cpx_timer_elapsed_ms(tm, ctypes.byref(v))
        if i: best = min(best, v.value)
    return best
d_out = DeviceBuf(B * L)
ONE = len(sys.argv) > 1
for name, gen in (("uniform +-8", lambda: (rs.rand(B, n) - 0.5) * 16.0),
                  ("gauss sigma 4 mean +-4", lambda: np.where(rs.rand(B, n) < 0.5, 4.0, -4.0) + rs.randn(B, n) * 2.8),
                  ("constant 1.0", lambda: np.ones((B, n)))):
    x = gen()
    d_in = DeviceBuf.from_array(x)
    if ONE:
        print(name, "cw %.3f ms" % timeit(d_in, d_out, "cw!", 2), flush=True)
        break
    print(name, "cw %.3f ms" % timeit(d_in, d_out, "cw!"), "wave %.3f ms" % timeit(d_in, d_out, "wave"), flush=True)
    d_in.free()
