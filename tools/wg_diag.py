"""Per-launch diagnosis of the fused cycle kernel (diagnostic build -DGSE_STAMPS=0): duration of the slowest workgroup,
the width of its ancestor range in tiles, how many tiles it expanded, and the guard-band evaluations of the launch."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_se_amd, bench
from gpu_se_amd import sim_base, _lib
n = 1 << 20
gpu_se_amd.seed(2)
_, _, _, pf = sim_base.get_parts(N_particles=n)
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 120
io = bench.make_inputs(cycles, 1.0)
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 8192)()
info = (ctypes.c_longlong * 8192)()
slow = ctypes.c_ulonglong(0)
prev_slow = 0
for c, (u, z) in enumerate(io):
    pf.predict(u, 1.0); pf.update(u, z); pf.resample()
    torch.cuda.synchronize()
    lib.gse_debug_block_times(buf, info)
    lib.gse_debug_slow_hits(ctypes.byref(slow))
    inf = np.array(info[:2048]).reshape(1024, 2)
    t = np.array(buf[:2048], dtype=np.float64).reshape(1024, 2)
    dur = (t[:, 1] - t[:, 0]) / 100.0
    total = (t[:, 1].max() - t[:, 0].min()) / 100.0
    width = inf[:, 0] & 0xFFFFF
    nact = inf[:, 0] >> 20
    worst = int(np.argmax(dur))
    print("cycle %3d  launch %.1f us  wg median %.1f max %.1f (width %d tiles, expanded %d)  widths: median %d p90 %d max %d  "
          "sum expanded %d  max expanded %d  wgs expanding >8: %d >24: %d  guard-band hits %d"
          % (c, total, np.median(dur), dur.max(), width[worst], nact[worst], np.median(width), np.percentile(width, 90), width.max(),
             nact.sum(), nact.max(), int(np.sum(nact > 8)), int(np.sum(nact > 24)), slow.value - prev_slow))
    prev_slow = slow.value
