"""Run sequences of the reference's open-loop benchmark (results/pf_openloop/pf_run_seq.py:14-131, :330-352 and
results/gsf_openloop/gsf_run_seq.py): per-method times of predict / update / resample against the number of particles
(or Gaussians), medians of N_runs calls.  Device classes: every call is followed by a synchronisation (the reference's
GPU timings are not), eager launches (filter.lazy = False: one kernel per method), plus the fused cycle as the classes run
it by default.  CPU column: the loop-faithful oracle (the reference's CPU algorithm), up to 2^12 particles.
usage: run_seq.py [pf|gsf] [max log2 N] [runs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_se_amd
from gpu_se_amd import sim_base
from oracle import gse_oracle as orc
import bench

which = sys.argv[1] if len(sys.argv) > 1 else "pf"
max_log = float(sys.argv[2]) if len(sys.argv) > 2 else (24 if which == "pf" else 18)
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 50


def timed(fn):
    gpu_se_amd.synchronize()
    t = time.perf_counter()
    fn()
    gpu_se_amd.synchronize()          # (polls the stream: gse_stream_synchronize)
    return time.perf_counter() - t


def cpu_pf(n, io):
    osp, omp = orc.bioreactor_noise()
    o = orc.ParticleFilterOracle(n, orc.MixtureOracle(*_prior()), osp, omp)
    out = {"predict": [], "update": [], "resample": []}
    for u, z in io[:3]:
        t = time.perf_counter(); o.predict(u, 1.0); out["predict"].append(time.perf_counter() - t)
        t = time.perf_counter(); o.update(u, z); out["update"].append(time.perf_counter() - t)
        t = time.perf_counter(); o.resample(); out["resample"].append(time.perf_counter() - t)
    return {k: float(np.median(v)) for k, v in out.items()}


def _prior():
    xss = orc.steady_state()
    return (np.zeros((2, 5)) + xss[None, :], np.array([np.diag([1e-4, 1e-7, 1e-3, 1e-3, 1e-7]), np.diag([1e-3, 1e-6, 1e-2, 1e-2, 1e-6])]),
            np.array([0.75, 0.25]))


print("# %s run sequences, medians of %d runs, microseconds; one MI355X" % (which, runs))
print("# %8s %10s %10s %10s %12s %12s | %s" % ("N", "predict", "update", "resample", "sum (eager)", "fused cycle",
                                              "CPU loop predict / update / resample, speed-up of the sum"))
io = bench.make_inputs(runs + 5, 1.0)
for lg in np.arange(1, max_log + 0.25, 1.0 if which == "pf" else 1.0):
    n = int(round(2 ** lg))
    gpu_se_amd.seed(3)
    _, _, _, f = sim_base.get_parts(N_particles=n, pf=(which == "pf"))
    f.lazy = False
    for u, z in io[:3]:
        f.predict(u, 1.0); f.update(u, z); f.resample()
    t = {"predict": [], "update": [], "resample": []}
    for u, z in io[3:3 + runs]:
        t["predict"].append(timed(lambda: f.predict(u, 1.0)))
        t["update"].append(timed(lambda: f.update(u, z)))
        t["resample"].append(timed(lambda: f.resample()))
    med = {k: 1e6 * float(np.median(v)) for k, v in t.items()}
    f.lazy = True
    cyc = []
    for u, z in io[3:3 + runs]:
        cyc.append(timed(lambda: (f.predict(u, 1.0), f.update(u, z), f.resample())))
    f.point_estimate()
    cpu = ""
    if which == "pf" and n <= 4096:
        c = cpu_pf(n, io)
        total = sum(c.values())
        cpu = "%9.0f %9.0f %9.0f  x%.0f" % (1e6 * c["predict"], 1e6 * c["update"], 1e6 * c["resample"],
                                            1e6 * total / sum(med.values()))
    print("  2^%-5g %10.1f %10.1f %10.1f %12.1f %12.1f | %s" % (lg, med["predict"], med["update"], med["resample"],
                                                                 sum(med.values()), 1e6 * float(np.median(cyc)), cpu))
