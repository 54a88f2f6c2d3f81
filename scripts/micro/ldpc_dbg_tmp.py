import ctypes, os, sys
import numpy as np
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/benchmarks")
from commpy_amd import _lib
from bench_kernels import Dev
from commpy_amd.channelcoding.ldpc import _device_code, get_ldpc_code_params
lib=_lib.load()
p = get_ldpc_code_params(os.path.join(ROOT, "commpy_amd/channelcoding/designs/ldpc/ieee80211n/1944.1296.txt"), True)
print("max vdeg", p["max_vnode_deg"], "max cdeg", p["max_cnode_deg"])
n,B=1944,16384
code=_device_code(p)
rs=np.random.RandomState(1)
llr=rs.randn(B,n)*2.0          # pure noise: nothing converges, every block runs all iterations
dev=Dev(lib)
d_llr=dev.put(llr); d_dec,d_out,d_it=dev.empty(B*n),dev.empty(B*n*8),dev.empty(B*4)
tm=ctypes.c_void_p(); lib.cpx_timer_create(ctypes.byref(tm))
for alg,name in ((1,"MSA"),(0,"SPA")):
    for dbg in (sys.argv[1:] or ("0","1","2","3")):
        os.environ["CPX_LDPC_DBG"]=dbg
        os.environ["CPX_LDPC_G"]="0"; os.environ["CPX_LDPC_THREADS"]="512"
        best=1e9
        for rep in range(3):
            lib.cpx_timer_start(tm,None)
            _lib.check(lib.cpx_ldpc_bp_decode_batch_dev(code,d_llr,B,alg,20,d_dec,d_out,d_it,None))
            lib.cpx_timer_stop(tm,None)
            v=ctypes.c_float(); lib.cpx_timer_elapsed_ms(tm,ctypes.byref(v)); best=min(best,v.value)
        its=dev.get(d_it,(B,),np.int32)
        print(name,"dbg",dbg,"%.3f ms"%best,"mean its",its.mean(), "-> %.1f M block-iterations/s"%(its.sum()/best/1e3))
