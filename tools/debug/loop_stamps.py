"""Phase times of the persistent GS-UKF closed loop (library built with GSE_CXXFLAGS=-DGSE_LOOP_STAMPS: rollout 0's moments rows
carry 100 MHz clock stamps instead of covariance entries).  usage: loop_stamps.py"""
import numpy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["GSE_RESIDENT"] = "1"; os.environ["GSE_GSF"] = "1"
sys.argv = ["closedloop_bench.py", "32", "6", "50", "0.1"]
import runpy
g = runpy.run_path(os.path.join(os.path.dirname(__file__), "..", "closedloop_bench.py"), run_name="__main__")
sim = g.get("sim") or g.get("s")
m = sim.moments[1:, 0, 6:12] * 10.0          # ns
d = numpy.diff(m, axis=1)
ctl = m[:, 2] > 0
names = ["predict", "update", "ancestors+gather", "moments", "control+rest"]
print("predict-only ticks (%d): predict %.2f us, moments+rest %.2f us, tick %.2f us" % (
    (~ctl).sum(), numpy.median(m[~ctl, 1] - m[~ctl, 0]) / 1e3, numpy.median(m[~ctl, 5] - m[~ctl, 1]) / 1e3,
    numpy.median(numpy.diff(m[:, 0])) / 1e3))
for i, n in enumerate(names):
    print("control ticks (%d): %s %.2f us" % (ctl.sum(), n, numpy.median(d[ctl, i]) / 1e3))
print("gap end of tick -> start of next: %.2f us" % (numpy.median(m[1:, 0] - m[:-1, 5]) / 1e3))
