"""rocprofv3 kernel trace of tools/tail_profile.py -> per-part statistics of the single-launch cycle kernel.
usage: tail_summary.py <dir with *_kernel_trace.csv>"""
import csv, glob, os, sys
import numpy as np
path = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"]
cyc = lambda r: "pf_step_kernel<true, true, false, true, 2, 3>" in name(r) or "pf_step_kernelILb1ELb1ELb0ELb1ELi2ELi3" in name(r)
mark = lambda r: "model_g" in name(r)
parts, cur = [], []
for r in rows:
    if mark(r):
        parts.append(cur); cur = []
    elif cyc(r):
        cur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
parts.append(cur)
print("kernel: pf_step_kernel<gather, predict, update, scan in+out> at N = 2^20; durations from %s" % os.path.basename(path))
if len(parts) > 1:
    d = np.array(parts[1])
    med = np.median(d)
    print("first %d cycles from the prior (the first has no resample to run): launches %d  median %.2f us  mean %.2f  p99 %.2f  max %.2f  max/median %.2f  launches > 2 x median: %d"
          % (len(d) + 1, len(d), med, d.mean(), np.percentile(d, 99), d.max(), d.max() / med, int(np.sum(d > 2 * med))))
    adv = [p[-1] for p in parts[2:] if len(p)]
    if adv:
        a = np.array(adv)
        print("adversarial patterns (12 patterns x 3 r, one single-launch cycle each): launches %d  median %.2f us  max %.2f  max / median of the first part %.2f"
              % (len(a), np.median(a), a.max(), a.max() / med))
        print("  per launch (us):", " ".join("%.1f" % v for v in a))
