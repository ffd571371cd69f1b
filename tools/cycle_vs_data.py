"""Per-cycle device time of the fused kernel against the structure of the weights (design data)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_se_amd, bench
from gpu_se_amd import sim_base
n = 1 << 20
gpu_se_amd.seed(1234)
_, _, _, pf = sim_base.get_parts(N_particles=n)
io = bench.make_inputs(80, 1.0)
rows = []
for c, (u, z) in enumerate(io):
    # time the fused launch of this cycle (gather of the previous resample + predict + update)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    pf.predict(u, 1.0)
    torch.cuda.synchronize()
    e0.record(); pf.update(u, z); e1.record()
    torch.cuda.synchronize()
    t_us = e0.elapsed_time(e1) * 1e3
    # structure of the weights that the NEXT launch will resample
    w = pf._w[0, :n].double()
    ess = float(w.sum() ** 2 / (w * w).sum())
    c_ = torch.cumsum(w, 0); c_ /= c_[-1].clone()
    r = 0.5
    thr = (torch.arange(n, device="cuda", dtype=torch.float64) + r) / n
    idx = torch.searchsorted(c_, thr).clamp(max=n - 1).reshape(-1, 1024)
    span = (idx[:, -1] - idx[:, 0] + 1).cpu().numpy()
    rows.append((c, t_us, ess, np.median(span), np.percentile(span, 90), span.max(), np.mean(span > 2048)))
    pf.resample(r=r)
print("cycle  kernel_us(this launch resamples the PREVIOUS row's weights)   ESS  span_med  span_p90  span_max  frac>2048")
for i, row in enumerate(rows):
    prev = rows[i - 1] if i else None
    if i >= 10 and i < 50:
        print("%3d  %6.1f   | prev weights: ESS %9.0f  med %6d  p90 %6d  max %7d  frac>2048 %.2f" % (row[0], row[1], prev[2], prev[3], prev[4], prev[5], prev[6]))
