import sys, json, subprocess, os
sys.path.insert(0, ".")
from gpu_se_amd import _lib
_lib.GSE_SCAN_MAX_TILES = int(sys.argv[1])
sys.argv = ["bench.py", "--log2-particles", sys.argv[2], "--steps", "100", "--warmup", "10", "--no-extras", "--no-cpu-baseline"]
import bench
d = bench.main()
print("max tiles", _lib.GSE_SCAN_MAX_TILES, "N=2^%s" % sys.argv[2], "value", d["value"], "fresh", d["roofline"]["device_us"], d["roofline"]["frac"], "steady", d["steady_state"]["device_us_per_cycle"], d["steady_state"]["frac"])
