import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, gpu_se_amd, bench
from gpu_se_amd.distributed import ShardedParticleFilter
n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
gpu_se_amd.seed(11)
a = ShardedParticleFilter.from_get_parts(n_local=n)
io = bench.make_inputs(40, 1.0)
for i in range(600):
    u, z = io[i % len(io)]
    a.predict(u, 1.0); a.update(u, z); a.resample()
    a.merge_begin(); a.merge_end()
torch.cuda.synchronize()
print(a.estimate_device.cpu().numpy())
