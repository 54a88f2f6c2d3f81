set -x
mkdir -p gpurun_out
ls /root/reference 2>&1 | head -2; nproc; lscpu | grep 'Model name'; rocminfo | grep -m2 gfx
python -m pytest tests/test_viterbi_gpu.py -m gpu -x -q 2>&1 | tail -15
python - <<'PY' 2>&1 | tee gpurun_out/first_bench.txt
import time, numpy as np, ctypes
from commpy_amd import _lib
from commpy_amd.channelcoding import Trellis, viterbi_decode, conv_encode
lib=_lib.load()
tr=Trellis(np.array([6]),np.array([[0o133,0o171]]))
rs=np.random.RandomState(0)
base=rs.randint(0,2,(64,1024)); coded=np.stack([conv_encode(m,tr) for m in base]).astype(float)
B=65536
llr=np.tile(8.0*coded-4.0,(B//64,1))+rs.randn(B,2060)*3.0
d_in=ctypes.c_void_p(); d_out=ctypes.c_void_p()
_lib.check(lib.cpx_malloc(ctypes.byref(d_in), llr.nbytes)); _lib.check(lib.cpx_malloc(ctypes.byref(d_out), B*1030))
_lib.check(lib.cpx_memcpy_h2d(d_in,_lib.ptr(llr),llr.nbytes))
tm=ctypes.c_void_p(); lib.cpx_timer_create(ctypes.byref(tm))
h=tr._device_handle()
for it in range(3):
    lib.cpx_timer_start(tm,None)
    _lib.check(lib.cpx_viterbi_decode_batch_dev(h,d_in,B,2060,1030,1035,30,1,d_out,None))
    lib.cpx_timer_stop(tm,None); ms=ctypes.c_float(); lib.cpx_timer_elapsed_ms(tm,ctypes.byref(ms))
    print('viterbi K=7 B=65536: %.3f ms -> %.2f Gbit/s info, %.1f GB/s algorithmic'%(ms.value, B*1024/ms.value/1e6, B*17510/ms.value/1e6))
PY
