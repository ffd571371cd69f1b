"""Builds libgse_hip.so (gfx950) from gpu_se_amd/csrc with hipcc.  In-tree, no JIT cache:
the .so travels to the GPU box with the repository snapshot."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libgse_hip.so")
SOURCES = ["gse_host.hip", "gse_pf.hip", "gse_gsukf.hip", "gse_rows.hip", "gse_mpc.hip", "gse_rtc.hip", "gse_loop.hip", "gse_pf_loop.hip"]
ARCH = "gfx950"
# per-source flags.  gse_loop.hip holds one kernel whose body is a loop over the ticks of a whole closed-loop run: with
# machine-level loop-invariant code motion every constant and address a tick needs is materialised before the loop and kept
# live through all of its phases, which costs 128 vector registers plus scratch; without, a tick allocates like one cycle.
EXTRA_FLAGS = {"gse_loop.hip": ["-mllvm", "-disable-machine-licm"]}


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "gse.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 and link the shared library."""
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
               "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-c", os.path.join(CSRC, src), "-o", obj]
        cmd[1:1] = os.environ.get("GSE_CXXFLAGS", "").split() + EXTRA_FLAGS.get(src, [])
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if out.strip() and verbose:
            print(out)
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out))
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
