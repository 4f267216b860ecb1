import sys, numpy as np
sys.path.insert(0, ".")
import riemann_amd as ra
from riemann_amd import proposals as P
from riemann_amd.workloads import logreg_data
N=65536
for n,d,its in ((10000,100,10),(10000,96,10),(10000,36,20),(1000,100,40)):
    X,y=logreg_data(n,d)
    m=ra.models.LogisticRegression(X,y)
    A=np.random.default_rng(d).standard_normal((d,d))/np.sqrt(d); M=A@A.T+np.eye(d)
    for name,mk in (("mala",lambda: P.MALA(0.02,m)),("mala_mass",lambda: P.VanillaHMC(0.02,1,m,M=M)),("hmc3_mass",lambda: P.VanillaHMC(0.02,3,m,M=M))):
        s=ra.Sampler(m,mk(),np.zeros((N,d)),seed=1)
        s.run(its,record=False); s.run(its,record=False)
        print("n=%d d=%d %-10s %.4f ms/it"%(n,d,name,s.last_run_ms/its),flush=True)
