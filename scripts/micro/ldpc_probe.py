"""Timing probe for the LDPC passes: fixed 8 iterations at an SNR where nothing converges (full working set, no
compaction).  The CPX_LDPC_DBG flag sweep belonged to ablation builds of csrc/ldpc.hip and is ignored by the shipped library."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
from commpy_amd import _lib
from bench_kernels import Dev
from commpy_amd.channelcoding.ldpc import _device_code, get_ldpc_code_params
lib = _lib.load()
p = get_ldpc_code_params(os.path.join(ROOT, "commpy_amd/channelcoding/designs/ldpc/ieee80211n/1944.1296.txt"), True)
n, E, B, IT = 1944, 7128, 32768, 8
rs = np.random.RandomState(1)
sigma = 1 / np.sqrt(10 ** (0.0 / 10.0) * (2.0 / 3) * 2)
llr = (2.0 * (1.0 + sigma * rs.randn(B, n)) / sigma ** 2)
dev = Dev(lib)
d_llr = dev.put(llr)
d_dec, d_out, d_it = dev.empty(B * n), dev.empty(B * n * 8), dev.empty(B * 4)
code = _device_code(p)
tm = ctypes.c_void_p(); lib.cpx_timer_create(ctypes.byref(tm))
for flags in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,1,2,4,8,3,10,11").split(",")]:
    os.environ["CPX_LDPC_DBG"] = str(flags)
    for alg in (1,):
        res = []
        for rep in range(3):
            lib.cpx_timer_start(tm, None)
            _lib.check(lib.cpx_ldpc_bp_decode_batch_dev(code, d_llr, B, alg, IT, d_dec, d_out, d_it, None))
            lib.cpx_timer_stop(tm, None)
            v = ctypes.c_float(); lib.cpx_timer_elapsed_ms(tm, ctypes.byref(v)); res.append(v.value)
        its = dev.get(d_it, (B,), np.int32)
        print("flags %2d alg %d: %.3f ms per decode of %d its -> %.3f ms/it (mean its %.2f)" % (flags, alg, min(res), IT, min(res) / IT, its.mean()), flush=True)
