"""Where does the per-step host time go when torch.distributed (nccl) is live?"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
mode = sys.argv[1]
if mode != "plain":
    import torch
    if mode in ("cuda", "dist"):
        torch.cuda.set_device(0)
        torch.zeros(1, device="cuda")
    if mode == "dist":
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
        dist.barrier()
import riemann_amd as ra
from riemann_amd import _lib
from riemann_amd.models import MultiGaussianDist
from riemann_amd.proposals import MetropolisRandomWalk
from riemann_amd.workloads import corr_gauss_cov
_lib.call("rmh_init", 0)
Ct = corr_gauss_cov(100)
th0 = np.random.default_rng(1).standard_normal((65536, 100))
m, p = MultiGaussianDist(np.zeros(100), Ct), MetropolisRandomWalk(0.0566 * Ct)
s = ra.Sampler(m, p, th0, seed=1)
p._sync()
hist = C.c_void_p()
_lib.call("rmh_history_create", s._state, 101, 100, C.byref(hist))
for it in range(4):
    t0 = time.perf_counter()
    _lib.call("rmh_run", m._h, p._h, s._state, 100, 0, 1, hist)
    t1 = time.perf_counter()
    ms = C.c_double(); _lib.call("rmh_last_run_ms", C.byref(ms), None)
    t2 = time.perf_counter()
    _lib.call("rmh_synchronize")
    t3 = time.perf_counter()
    print(mode, "launch %.2f ms  event-wait %.2f ms  sync %.2f ms  kernel %.2f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, ms.value), flush=True)
if mode == "dist":
    dist.destroy_process_group()
