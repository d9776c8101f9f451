import sys, ctypes as C, numpy as np
sys.path[:0]=["/root/repo","/root/repo/tests"]
import lis_amd
from lis_amd import check, DeviceArray as DA
lib=lis_amd.load()
n=1<<27
src,dst=DA(13*n,np.float64),DA(n,np.float64)
check(lib.liship_memset(src.ptr,0,src.nbytes,None))
t=C.c_void_p(); check(lib.liship_timer_create(C.byref(t))); ms=C.c_float()
for reads in (13,8,1):
  for w in (0,2,4,8,16,32):
    for _ in range(5): check(lib.liship_stream_yardstick(reads,n,src.ptr,dst.ptr,w,None))
    check(lib.liship_timer_start(t,None))
    for _ in range(20): check(lib.liship_stream_yardstick(reads,n,src.ptr,dst.ptr,w,None))
    check(lib.liship_timer_stop(t,None)); check(lib.liship_device_synchronize()); check(lib.liship_timer_elapsed_ms(t,C.byref(ms)))
    print("yardstick reads",reads,"wgs/CU",w,"GB/s %.0f"%((reads+1)*8*n/(ms.value/20*1e-3)/1e9),flush=True)
