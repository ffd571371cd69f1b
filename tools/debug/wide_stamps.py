"""Phase timeline of a workgroup that takes the LOOK-UP path of the gather stage (wide ancestor ranges), from s_memtime stamps.
Needs the diagnostic patch next to this file and a diagnostic build, neither of which is part of the product sources:
    git apply tools/debug/wide_stamps.patch && GSE_CXXFLAGS=-DGSE_STAMPS=0 python -m gpu_se_amd.build --force
    python tools/debug/wide_stamps.py ; git checkout gpu_se_amd/csrc/gse_pf_kernels.h
Round 3, two adversarial patterns at 2^20 (ticks of ~2.1 GHz): start -> wide branch 7-8 k, thresholds 7-8 k (fp64 divisions,
the SIMDs shared with three normal workgroups), tile search 6-7 k, quads 16-18 k, windows 8-10 k, the rest of the cycle 19-20 k."""
import ctypes, os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import gpu_se_amd, bench
from gpu_se_amd import sim_base, _lib
from gpu_util import adversarial_weights
n = 1 << 20
gpu_se_amd.seed(1234)
_, _, _, pf = sim_base.get_parts(N_particles=n)
lib = ctypes.CDLL(_lib.LIB_PATH)
bench.device_warm_up(torch, n, 1.0)
xss = np.array([1.55555556, 26.01626016, 5.2711538, 0.0, 15.18857143], dtype=np.float32)
u, z = [0.06, 0.2], np.array([280.0, 611.4538406938659])
rng = np.random.default_rng(n)
pats = adversarial_weights(n, rng)
names = {14: "wg start", 15: "wide branch", 16: "draw done", 22: "taus done", 17: "tiles found", 18: "quads found", 19: "windows done", 20: "wg end"}
for name in ("light_half", "eight_steps_down"):
    w = pats[name]
    for rep in range(3):
        pf.particles = np.repeat(xss[None, :], n, axis=0)
        pf.weights = (w / np.nansum(w)).astype(np.float32)
        pf.update(u, z); pf.resample(r=0.37); pf.predict(u, 0.1); pf.update(u, z)
        torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 32)()
    lib.gse_debug_stamps(buf)
    t = {i: int(buf[i]) for i in names}
    print(name, "span", int(buf[21]) & 0xfffff, "listed", int(buf[21]) >> 20)
    prev = t[14]
    for i in (14, 15, 16, 22, 17, 18, 19, 20):
        print("   %-14s +%8d ticks (phase %7d)" % (names[i], t[i] - t[14], t[i] - prev))
        prev = t[i]
