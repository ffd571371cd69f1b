"""Phase timeline of the fused cycle kernel from in-kernel s_memtime stamps (diagnostic build -DGSE_STAMPS=<block>)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_se_amd, bench
from gpu_se_amd import sim_base, _lib
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log2n
gpu_se_amd.seed(1)
_, _, _, pf = sim_base.get_parts(N_particles=n)
io = bench.make_inputs(60, 1.0)
lib = ctypes.CDLL(_lib.LIB_PATH)
names = ["start", "draw@0", "prefix", "marks done", "ancestors done", "loads issued", "draw@3", "rows landed", "ODE done",
         "update done", "scan-out done", "tiles found"]
order = [0, 1, 2, 11, 3, 4, 5, 6, 7, 8, 9, 10]
acc = []
for c, (u, z) in enumerate(io):
    pf.predict(u, 1.0); pf.update(u, z); pf.resample()
    if c >= 20:
        buf = (ctypes.c_ulonglong * 32)()
        lib.gse_debug_stamps(buf)
        acc.append(np.array(buf[:12], dtype=np.float64))
a = np.median(np.array(acc), axis=0)
d = a - a[0]
prev = 0.0
for i in order:
    print("%-16s +%9.0f ticks   (phase %8.0f)" % (names[i], d[i], d[i] - prev))
    prev = d[i]
