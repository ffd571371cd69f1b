"""Entry / exit time of every workgroup of the fused cycle kernel (diagnostic build -DGSE_STAMPS=0): is the launch one
generation of workgroups or several?"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_se_amd, bench
from gpu_se_amd import sim_base, _lib
n = 1 << 20
gpu_se_amd.seed(1)
_, _, _, pf = sim_base.get_parts(N_particles=n)
io = bench.make_inputs(40, 1.0)
lib = ctypes.CDLL(_lib.LIB_PATH)
for c, (u, z) in enumerate(io):
    pf.predict(u, 1.0); pf.update(u, z); pf.resample()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8192)()
info = (ctypes.c_longlong * 8192)()
lib.gse_debug_block_times(buf, info)
inf = np.array(info[:2048]).reshape(1024, 2)
t = np.array(buf[:2048], dtype=np.float64).reshape(1024, 2)
t0 = t[:, 0].min()
start = (t[:, 0] - t0) / 100.0      # us (100 MHz)
end = (t[:, 1] - t0) / 100.0
print("workgroups: 1024; start us: min %.2f  p25 %.2f  median %.2f  p75 %.2f  max %.2f" % (start.min(), *np.percentile(start, [25, 50, 75]), start.max()))
print("            end   us: min %.2f  median %.2f  max %.2f ; duration median %.2f  p90 %.2f max %.2f" % (end.min(), np.median(end), end.max(), np.median(end - start), np.percentile(end - start, 90), (end - start).max()))
hist, edges = np.histogram(start, bins=12)
print("start histogram:", list(zip(np.round(edges[:-1], 1), hist)))

dur = end - start
for pos in (0, 3):
    sel = inf[:, 1] == pos
    print("draw position %d: %4d workgroups, duration median %.2f p90 %.2f" % (pos, sel.sum(), np.median(dur[sel]), np.percentile(dur[sel], 90)))
for lo, hi in ((0, 1024), (1024, 2048), (2048, 8192), (8192, 1 << 40)):
    sel = (inf[:, 0] > lo) & (inf[:, 0] <= hi)
    if sel.any():
        print("span in (%d, %d]: %4d workgroups, duration median %.2f p90 %.2f max %.2f" % (lo, hi, sel.sum(), np.median(dur[sel]), np.percentile(dur[sel], 90), dur[sel].max()))
bx = np.arange(1024)
for x in range(8):
    sel = (bx % 8) == x
    print("blockIdx %% 8 == %d (one XCD): duration median %.2f" % (x, np.median(dur[sel])), end=";  ")
print()
print("end-time histogram (us):", list(zip(np.round(np.histogram(end, bins=14)[1][:-1], 1), np.histogram(end, bins=14)[0])))
# groups of 4 consecutive-in-XCD workgroups: bx, bx+8, bx+16, bx+24 share an XCD; a CU hosts 4 workgroups
order = np.argsort(end)
print("last 12 to finish: bx", order[-12:], "end", np.round(end[order[-12:]], 1), "dur", np.round(dur[order[-12:]], 1), "span", inf[order[-12:], 0])
