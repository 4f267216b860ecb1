import time, sys, numpy as np, ctypes as C
sys.path.insert(0,'.')
import riemann_amd as ra
from riemann_amd import _lib
d=100; chains=65536; its=200
s=ra.Sampler(ra.models.MultiGaussianDist(np.zeros(d),np.eye(d)), ra.proposals.MetropolisRandomWalk(0.05*np.eye(d)), np.zeros((chains,d)), seed=1)
s.run(its)
n=its+1
pool=_lib.PinnedPool()
th=pool.empty((n,chains,d)); del th
for rep in range(2):
    t0=time.perf_counter(); th=pool.empty((n,chains,d)); t1=time.perf_counter()
    lp=np.empty((n,chains)); acc=np.empty((its,chains),dtype=np.uint8); t2=time.perf_counter()
    _lib.call("rmh_history_download", s._hist, 0, n, th.ctypes.data, lp.ctypes.data); t3=time.perf_counter()
    _lib.call("rmh_history_accept", s._hist, 0, its, acc); t4=time.perf_counter()
    print("pool.empty %.1f ms, np.empty %.1f ms, download(th+lp) %.1f ms, accept %.1f ms"%((t1-t0)*1e3,(t2-t1)*1e3,(t3-t2)*1e3,(t4-t3)*1e3))
    del th
