"""ITSE statistics of the Philox-driven resident GS-UKF closed loop (64 Gaussians, control every tick, end_time = 50: G11's
gsf64_c01) over many rollouts, against the reference's own runs (tests/golden/loop_envelope_gsf.npz + loop_envelope.npz).
usage: gsf_loop_stats.py [batches = 32] [base seed = 5000] [out.npz]     (32 rollouts per batch, distinct seeds per batch)
Prints median / mean / sd / quantiles of both samples, Mann-Whitney U and Kolmogorov-Smirnov p-values, per-window RMS error
ratios, and the correlation of the plant-noise realisations between rollouts of one batch (independence check)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpu_se_amd
from gpu_se_amd import sim_base

batches = int(sys.argv[1]) if len(sys.argv) > 1 else 32
base = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
out = sys.argv[3] if len(sys.argv) > 3 else None
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
perf, errs, corr = [], [], []
win_sq = np.zeros((5, 2))
for i in range(batches):
    gpu_se_amd.seed(base + i)
    sim = sim_base.BatchedSimulation(batch=32, N_particles=64, dt_control=0.1, dt_predict=0.1, end_time=50, pf=False, resident=True,
                                     seed=base + i)
    sim.simulate()
    assert sim.mpc_frac == 1.0
    perf.append(np.asarray(sim.performance))
    e = sim.ys_f - sim.ys[:, :, [0, 2]]
    errs.append(np.sqrt(np.mean(e[1:] ** 2, axis=0)))             # (B, 2) RMS error per rollout
    for k, lo in enumerate(range(0, 500, 100)):
        win_sq[k] += np.sum(e[max(lo, 1):lo + 100] ** 2, axis=(0, 1))
    # independence of the rollouts of one batch: the plant's state-noise realisations (increments of xs beyond the model) differ
    dx = np.diff(sim.xs[:, :, 2], axis=0)                         # (ticks - 1, B)
    c = np.corrcoef((dx - dx.mean(axis=1, keepdims=True)).T)
    corr.append(c[np.triu_indices(32, 1)])
perf = np.concatenate(perf)
corr = np.concatenate(corr)
print("device: %d rollouts, ITSE median %.4g mean %.4g sd %.4g q10 %.4g q90 %.4g" % (
    len(perf), np.median(perf), perf.mean(), perf.std(), *np.quantile(perf, [0.1, 0.9])))
bm = np.array([np.median(perf[32 * i:32 * (i + 1)]) for i in range(batches)])
print("  medians of the %d batches of 32: min %.4g max %.4g sd %.4g" % (batches, bm.min(), bm.max(), bm.std()))
print("  correlation between rollouts of a batch (fatty-acid increments about the batch mean): mean %.4f, max |.| %.4f "
      "(independent streams: -1/31 = %.4f expected after removing the batch mean)" % (corr.mean(), np.abs(corr).max(), -1 / 31))
ref = []
for name in ("loop_envelope_gsf.npz", "loop_envelope.npz"):
    path = os.path.join(ROOT, "tests", "golden", name)
    if os.path.exists(path):
        g = np.load(path)
        ref.append(g["gsf64_c01_performance"])
        print("reference %s: %d runs, median %.4g mean %.4g sd %.4g" % (name, len(ref[-1]), np.median(ref[-1]), ref[-1].mean(),
                                                                        ref[-1].std()))
if ref:
    from scipy import stats
    r = np.concatenate(ref)
    print("reference pooled: %d runs, median %.4g mean %.4g sd %.4g q10 %.4g q90 %.4g" % (
        len(r), np.median(r), r.mean(), r.std(), *np.quantile(r, [0.1, 0.9])))
    print("Mann-Whitney U p = %.4g, Kolmogorov-Smirnov p = %.4g, ratio of medians %.4f" % (
        stats.mannwhitneyu(perf, r, alternative="two-sided").pvalue, stats.ks_2samp(perf, r).pvalue, np.median(perf) / np.median(r)))
path = os.path.join(ROOT, "tests", "golden", "loop_envelope_gsf.npz")
if os.path.exists(path):
    er = np.load(path)["gsf64_c01_err"].astype(np.float64)        # (seeds, ticks, 2)
    for k, lo in enumerate(range(0, 500, 100)):
        sl = slice(max(lo, 1), lo + 100)
        rms_r = np.sqrt(np.mean(er[:, sl] ** 2, axis=(0, 1)))
        rms_d = np.sqrt(win_sq[k] / (len(perf) * (sl.stop - sl.start)))
        print("  ticks %3d..%3d: RMS estimation error device %s reference %s ratio %s" % (sl.start, sl.stop, rms_d, rms_r, rms_d / rms_r))
if out:
    np.savez(out, performance=perf, rms=np.concatenate(errs), batch_medians=bm)
