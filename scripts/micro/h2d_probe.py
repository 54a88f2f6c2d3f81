import ctypes, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from commpy_amd import _lib
lib = _lib.load(); _lib.require_device()
n = 65536 * 2060
x = np.random.RandomState(0).randn(n)
p = ctypes.c_void_p(); _lib.check(lib.cpx_malloc(ctypes.byref(p), x.nbytes))
for chunks in (1, 2, 4, 8, 1, 4):
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter()
        step = n // chunks
        for c in range(chunks):
            _lib.check(lib.cpx_memcpy_h2d(ctypes.c_void_p(p.value + c * step * 8), ctypes.c_void_p(x.ctypes.data + c * step * 8), step * 8))
        best = min(best, time.perf_counter() - t0)
    print("H2D 1.08 GB pageable in %d chunk(s): %.2f ms = %.1f GB/s" % (chunks, best * 1e3, x.nbytes / best / 1e9), flush=True)
