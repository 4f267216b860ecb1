import time, ctypes as C, numpy as np, sys
sys.path.insert(0,'.')
import riemann_amd as ra
from riemann_amd import _lib
d=100; chains=65536; its=200
s=ra.Sampler(ra.models.MultiGaussianDist(np.zeros(d),np.eye(d)), ra.proposals.MetropolisRandomWalk(0.05*np.eye(d)), np.zeros((chains,d)), seed=1)
s.run(its)
n=its+1
pool=_lib.PinnedPool()
t0=time.perf_counter(); th=pool.empty((n,chains,d)); t1=time.perf_counter()
print("pinned alloc %.3f s"%(t1-t0), type(th.base))
for rep in range(3):
    t0=time.perf_counter(); _lib.call("rmh_history_download", s._hist, 0, n, th.ctypes.data, None); t1=time.perf_counter()
    print("download thetas only: %.1f ms = %.1f GB/s"%((t1-t0)*1e3, th.nbytes/(t1-t0)/1e9))
lp=np.empty((n,chains))
t0=time.perf_counter(); _lib.call("rmh_history_download", s._hist, 0, n, None, lp.ctypes.data); t1=time.perf_counter()
print("logpost %.1f ms"%((t1-t0)*1e3))
acc=np.empty((its,chains),dtype=np.uint8)
t0=time.perf_counter(); _lib.call("rmh_history_accept", s._hist, 0, its, acc); t1=time.perf_counter()
print("accept %.1f ms"%((t1-t0)*1e3))
th2=np.empty((n,chains,d))
t0=time.perf_counter(); _lib.call("rmh_history_download", s._hist, 0, n, th2.ctypes.data, None); t1=time.perf_counter()
print("pageable thetas: %.1f GB/s"%(th2.nbytes/(t1-t0)/1e9))
