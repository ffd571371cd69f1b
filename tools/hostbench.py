# micro-benchmark of the host side of one cycle (run on the GPU box)
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import gpu_se_amd
from gpu_se_amd import sim_base
_,_,_,pf = sim_base.get_parts(N_particles=1024)
u=[0.05,0.1]; z=np.array([280.,611.])
for name,fn in (("predict",lambda: pf.predict(u,1.0)),("update",lambda: pf.update(u,z)),("resample",lambda: pf.resample())):
    for _ in range(200): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(2000): fn()
    dt=(time.perf_counter()-t)/2000; torch.cuda.synchronize()
    print(name, "host us/call %.2f"%(dt*1e6))
