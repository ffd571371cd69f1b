"""Phase timeline of the fused merge launch (resampled moments + tail) from in-kernel s_memtime stamps
(diagnostic build: GSE_CXXFLAGS=-DGSE_STAMPS=<workgroup> python gpu_se_amd/build.py --force)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import gpu_se_amd, bench
from gpu_se_amd import _lib
from gpu_se_amd.distributed import ShardedParticleFilter
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
gpu_se_amd.seed(11)
a = ShardedParticleFilter.from_get_parts(n_local=n)
io = bench.make_inputs(60, 1.0)
lib = ctypes.CDLL(_lib.LIB_PATH)
rows = []
for c, (u, z) in enumerate(io):
    a.predict(u, 1.0); a.update(u, z); a.resample(); a.merge_begin(); a.merge_end()
    if c >= 20:
        buf = (ctypes.c_ulonglong * 32)()
        lib.gse_debug_mstamps(buf)
        rows.append(np.array(buf[:17], dtype=np.float64))
r = np.array(rows)
names = {0: "start", 1: "tile prefix built", 2: "counted + listed", 3: "list worked off", 4: "block sums + stores issued",
         5: "stores acknowledged", 6: "ticket drawn", 8: "(last wg) ticket", 9: "(last wg) partials read+added", 10: "(last wg) moments in LDS",
         11: "(last wg) tail done"}
med = np.median(r, axis=0)
print("list length of the stamped wavefront: median %.0f  (min %.0f, max %.0f)" % (med[16], r[:, 16].min(), r[:, 16].max()))
for seq in ((0, 1, 2, 3, 4, 5, 6), (8, 9, 10, 11)):
    prev = None
    for i in seq:
        d = np.median(r[:, i] - r[:, seq[0]])
        print("%-34s +%8.0f ticks (phase %7.0f)" % (names[i], d, d - (prev or 0)))
        prev = d
print("(s_memtime ticks; the launch takes ~16.5 us for ~34 000 ticks: 1 tick ~ 0.48 ns)")
