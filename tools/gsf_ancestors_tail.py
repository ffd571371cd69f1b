"""Worst cases of the large-bank ancestors kernel, for `rocprofv3 --kernel-trace`: every adversarial weight pattern of
tests/gpu_util.py at 2^18 Gaussians through the deferred resample (gse_gsukf_resample_begin -> ancestors_scatter_kernel, or
ancestors_kernel with GSE_ANCESTORS_BY_SEARCH=1).  With a trace directory as argument it prints the kernel's duration per
pattern instead (launch order = the order printed here)."""
import os, sys, glob, csv
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from gpu_util import adversarial_weights
n = 1 << 18
names = [(name, r) for name in adversarial_weights(16, np.random.default_rng(0)) for r in (0.0, 0.37)]
if len(sys.argv) > 1:
    rows = [r for f in glob.glob(os.path.join(sys.argv[1], "*", "*kernel_trace.csv")) for r in csv.DictReader(open(f))]
    rows = sorted((r for r in rows if "ancestors" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows][-len(names):]
    print("# %s at 2^18 Gaussians, one launch per pattern, microseconds" % rows[-1]["Kernel_Name"].split("(")[0])
    for (name, r), d in zip(names, durs):
        print("%-22s r=%-5g %7.2f" % (name, r, d))
    print("median %.2f  max %.2f" % (float(np.median(durs)), max(durs)))
    sys.exit(0)
import torch
import gpu_se_amd
from gpu_se_amd import sim_base
gpu_se_amd.seed(5)
_, _, _, gf = sim_base.get_parts(N_particles=n, pf=False)
rng = np.random.default_rng(n)
for _ in range(3):
    gf.weights = rng.random(n).astype(np.float32)
    gf.resample(r=0.5)
torch.cuda.synchronize()
for name, w in adversarial_weights(n, rng).items():
    for r in (0.0, 0.37):
        gf.weights = w.astype(np.float32)
        gf.resample(r=r)
        torch.cuda.synchronize()
        print(name, r, flush=True)
