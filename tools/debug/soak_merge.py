"""Soak of round 5's new paths (one MI355X): thousands of cycles with wandering inputs, measurement jumps of N(0, 150) mg/L in
2-3 % of the cycles (the weight scale engages) and random r.
 (1) the merge WITHOUT a flush (gse_pf_merge_begin: resampled moments + the launch's last workgroup) every cycle, against the
     flushed merge of a twin filter every 20 cycles and at every jump; the two filters' states stay bit-identical throughout;
 (2) ShardedParticleFilter.run_cycles in batches of 7 against the three methods;
 (3) the GS-UKF's run_cycles in its three regimes against the three methods.
usage: soak_merge.py [cycles = 3000] [seed = 1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import gpu_se_amd
from gpu_se_amd import sim_base
from gpu_se_amd.distributed import ShardedParticleFilter
from gpu_se_amd.filter import ParallelGaussianSumUnscentedKalmanFilter as GSF
from gpu_se_amd.model import Bioreactor

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
xss = Bioreactor.find_SS(np.array([0.06, 0.2]), np.array([260 / 180, 640 / 24.6, 1000 / 116, 0, 0]))
z0 = np.array(Bioreactor.static_outputs(xss, None))


def inputs(rng, k):
    us = np.stack([rng.uniform(0.03, 0.09, k), rng.uniform(0.1, 0.3, k)], axis=-1)
    zs = z0[None, :] + rng.normal(0, 0.4, (k, 2))
    jump = rng.random(k) < 0.025
    zs[jump, 0] += rng.normal(0, 150, jump.sum())       # (far enough for the weight scale, not for fp64's own range)
    return us, zs, jump


for n in (1 << 16, 70001, 1 << 20):
    rng = np.random.default_rng(seed)
    gpu_se_amd.seed(100 + seed)
    a = ShardedParticleFilter.from_get_parts(n_local=n)
    gpu_se_amd.seed(100 + seed)
    b = ShardedParticleFilter.from_get_parts(n_local=n)
    us, zs, jump = inputs(rng, cycles)
    worst, checks, dead = 0.0, 0, 0
    for k in range(cycles):
        r = float(rng.random())
        for f in (a, b):
            f.predict(us[k], 0.1); f.update(us[k], zs[k]); f.resample(r=r)
        a.merge_begin(); ma = a.merge_end()
        if k % 20 == 0 or jump[k]:
            mnew = ma.cpu().numpy().copy()
            b.merge_begin(flushed=True); mold = b.merge_end().cpu().numpy()
            sa, sb = a._dev_status.item(), b._dev_status.item()
            assert sa == sb and sa in (0, 1), (n, k, sa, sb)
            if sa == 1:                 # (a jump so far out that every likelihood underflowed in fp64 too: no posterior to compare;
                dead += 1               #  the resample then copies particle 0, on both routes alike -- checked below)
                continue
            err = np.max(np.abs(mnew[1:6] - mold[1:6]) / (np.abs(mold[1:6]) + 1e-300))
            err2 = np.max(np.abs(mnew[6:21] - mold[6:21])) / np.abs(mold[6:21]).max()
            worst = max(worst, err, err2 * 1e-3)
            assert err < 1e-11 and err2 < 1e-9, (n, k, err, err2)
            checks += 1
        if k % 250 == 249:
            a.local._flush(); b.local._flush()
            assert torch.equal(a.local._x, b.local._x) and torch.equal(a.local._w, b.local._w) and torch.equal(a.local._island, b.local._island), (n, k)
    a.local._check_status()
    print("merge without a flush N=%d: %d cycles, %d jumps, %d comparisons with the flushed merge, worst relative difference of the "
          "mean %.2e, %d jumps beyond every likelihood's range (no weight: status 1 on both routes): ok"
          % (n, cycles, int(jump.sum()), checks, worst, dead), flush=True)

for n in (1 << 15, 1 << 20):
    rng = np.random.default_rng(seed + 10)
    gpu_se_amd.seed(200 + seed)
    a = ShardedParticleFilter.from_get_parts(n_local=n)
    gpu_se_amd.seed(200 + seed)
    b = ShardedParticleFilter.from_get_parts(n_local=n)
    total = (cycles // 7) * 7
    us, zs, jump = inputs(rng, total)
    for k0 in range(0, total, 7):
        a.run_cycles(us[k0:k0 + 7], zs[k0:k0 + 7], 0.1)
        for k in range(k0, k0 + 7):
            b.predict(us[k], 0.1); b.update(us[k], zs[k]); b.resample()
        if (k0 // 7) % 20 == 0:
            a.merge_begin(); b.merge_begin()
            assert torch.equal(a.merge_end(), b.merge_end()), (n, k0)
    a.local._flush(); b.local._flush()
    assert torch.equal(a.local._x, b.local._x) and torch.equal(a.local._w, b.local._w) and torch.equal(a.local._island, b.local._island)
    print("sharded run_cycles N=%d: %d cycles in batches of 7, %d jumps: the three methods' ensemble bit for bit: ok" % (n, total, int(jump.sum())), flush=True)

for n in (64, 5000, 1 << 16, 1 << 18):
    rng = np.random.default_rng(seed + 20)
    banks = []
    for _ in range(2):
        gpu_se_amd.seed(300 + seed)
        _, _, _, gf = sim_base.get_parts(N_particles=n, pf=False)
        banks.append(gf)
    a, b = banks
    total = (min(cycles, 1500) // 5) * 5
    us, zs, _ = inputs(rng, total)
    zs = z0[None, :] + rng.normal(0, 0.4, (total, 2))
    for k0 in range(0, total, 5):
        a.run_cycles(us[k0:k0 + 5], zs[k0:k0 + 5], 0.1)
        for k in range(k0, k0 + 5):
            b.predict(us[k], 0.1); b.update(us[k], zs[k]); b.resample()
    assert torch.equal(a.means.as_subclass(torch.Tensor), b.means.as_subclass(torch.Tensor))
    assert torch.equal(a.covariances.as_subclass(torch.Tensor), b.covariances.as_subclass(torch.Tensor))
    assert int(a._status.item()) == 0
    print("GS-UKF run_cycles N=%d: %d cycles in batches of 5: the three methods' bank bit for bit: ok" % (n, total), flush=True)
