"""Repository rules that can be checked mechanically (CPU)."""
import os
import re

from tests.conftest import ROOT


def _py_files(top):
    for d, _, files in os.walk(os.path.join(ROOT, top)):
        for f in files:
            if f.endswith(".py"):
                yield os.path.join(d, f)


def test_product_never_touches_the_oracle_or_the_reference():
    """The shipped package must not import, read or mention a path into oracle/ or /root/reference (the oracle is
    test infrastructure; the reference does not exist on the GPU box)."""
    for path in _py_files("gpu_se_amd"):
        text = open(path).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), path
        assert "/root/reference" not in text, path
    for path in [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")] + list(_py_files("tests")):
        if path.endswith(os.path.join("golden", "make_golden.py")) or path.endswith("test_hygiene.py"):
            continue                                   # the fixture generator runs in the build container only
        assert "/root/reference" not in open(path).read(), path


def test_no_cpu_fallback_in_the_classes():
    from gpu_se_amd import _lib
    import gpu_se_amd.filter.particle as particle
    src = open(particle.__file__).read()
    assert "numpy.cumsum" not in src and "searchsorted" not in src
    assert os.path.basename(_lib.LIB_PATH) == "libgse_hip.so"
