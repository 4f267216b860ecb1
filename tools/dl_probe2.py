import time, sys, numpy as np
sys.path.insert(0,'.')
import riemann_amd as ra
d=100; chains=65536; its=200
s=ra.Sampler(ra.models.MultiGaussianDist(np.zeros(d),np.eye(d)), ra.proposals.MetropolisRandomWalk(0.05*np.eye(d)), np.zeros((chains,d)), seed=1)
for rep in range(3):
    s.run(its)
    t0=time.perf_counter(); s._materialise(); t1=time.perf_counter()
    a=s._host_thetas
    b=a
    while isinstance(b, np.ndarray) and b.base is not None: b=b.base
    print("materialise %.1f ms = %.1f GB/s; base %s; pool bufs %s"%((t1-t0)*1e3, a.nbytes/(t1-t0)/1e9, type(b).__name__, [x.nbytes for x in s._pool._allocs] if s._pool else None))
    del a, b
