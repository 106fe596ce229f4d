import time, numpy as np, torch, sys
sys.path.insert(0,'.')
from gpupathtracer_amd import api
n=1920*1080*3*4
d=api.device_alloc(n)
for rep in range(3):
    t=time.perf_counter(); 
    for i in range(20): out=api.device_download(d,n)
    print("pageable hipMemcpy D2H (new numpy buffer each time): %.2f ms"%((time.perf_counter()-t)/20*1e3))
import ctypes
from gpupathtracer_amd import _lib
buf=np.empty(n,np.uint8)
L=_lib.lib()
for rep in range(3):
    t=time.perf_counter()
    for i in range(20): L.gpt_device_download(buf.ctypes.data_as(ctypes.c_void_p), d, n)
    print("pageable hipMemcpy D2H (same buffer): %.2f ms"%((time.perf_counter()-t)/20*1e3))
x=torch.empty(n,dtype=torch.uint8,device='cuda'); p=torch.empty(n,dtype=torch.uint8).pin_memory()
torch.cuda.synchronize()
for rep in range(3):
    t=time.perf_counter()
    for i in range(20): p.copy_(x,non_blocking=True); torch.cuda.synchronize()
    print("pinned D2H: %.2f ms"%((time.perf_counter()-t)/20*1e3))
q=np.empty(n,np.uint8)
for rep in range(3):
    t=time.perf_counter()
    for i in range(20): np.copyto(q, p.numpy())
    print("host memcpy 24.9MB: %.2f ms"%((time.perf_counter()-t)/20*1e3))
