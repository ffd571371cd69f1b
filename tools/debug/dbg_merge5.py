"""Round-5 development check: the no-flush merge of ShardedParticleFilter against the flushed one, and its cost."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import gpu_se_amd
from gpu_se_amd import _lib
from gpu_se_amd.distributed import ShardedParticleFilter
import bench

def run(n):
    gpu_se_amd.seed(11)
    a = ShardedParticleFilter.from_get_parts(n_local=n)
    gpu_se_amd.seed(11)
    b = ShardedParticleFilter.from_get_parts(n_local=n)
    io = bench.make_inputs(40, 1.0)
    for i, (u, z) in enumerate(io[:12]):
        for f in (a, b):
            f.predict(u, 1.0); f.update(u, z); f.resample(r=0.37 + 0.01 * i)
        a.merge_begin(); m_new = a.merge_end().clone(); e_new = a.estimate_device.clone(); row_new = a._dev_rows.clone()
        assert a.local._pend_gather, "the merge flushed"
    # b: flushed route
    b.merge_begin(flushed=True); m_old = b.merge_end().clone(); e_old = b.estimate_device.clone(); row_old = b._dev_rows.clone()
    torch.cuda.synchronize()
    print("n", n, "status", a._dev_status.item(), b._dev_status.item())
    print("row logA", row_new[0, 0].item(), row_old[0, 0].item(), "S0", row_new[0, 1].item(), row_old[0, 1].item())
    mn, mo = m_new.cpu().numpy(), m_old.cpu().numpy()
    print("mean rel diff", np.max(np.abs(mn[1:6] - mo[1:6]) / (np.abs(mo[1:6]) + 1e-300)))
    print("S2 rel diff", np.max(np.abs(mn[6:21] - mo[6:21]) / (np.abs(mo[6:21]).max())))
    print("est new", e_new.cpu().numpy()); print("est old", e_old.cpu().numpy())
    # against the one-thread estimates kernel
    ref = torch.zeros(6, dtype=torch.float64, device="cuda")
    _lib.call("gse_moments_estimates", m_new.data_ptr(), 1, ref.data_ptr(), torch.cuda.current_stream().cuda_stream)
    print("est ref", ref.cpu().numpy())
    # the state of a was not touched: particles equal b's after both flush
    a.local._flush()
    assert torch.equal(a.local._x, b.local._x) and torch.equal(a.local._w, b.local._w)
    print("state untouched ok")
    # timing
    for mode in ("plain", "estimate"):
        def cyc(i):
            u, z = io[i % len(io)]
            a.predict(u, 1.0); a.update(u, z); a.resample()
            if mode == "estimate":
                a.merge_begin(); a.merge_end()
        for i in range(50): cyc(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for i in range(400): cyc(i)
        e1.record(); torch.cuda.synchronize()
        print(mode, "device us/cycle", e0.elapsed_time(e1) * 1e3 / 400, "wall us/cycle", (time.perf_counter() - t0) * 1e6 / 400)

for n in (1 << 16, 1 << 20):
    run(n)
