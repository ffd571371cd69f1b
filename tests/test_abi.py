"""CPU checks of the C-ABI library and the host-side logic (no kernel is launched here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from tests.conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as entry
    entry.build()
    from gpu_se_amd import _lib
    return _lib


def test_every_declared_symbol_is_exported(lib):
    header = open(os.path.join(ROOT, "include", "gse.h")).read()
    declared = set(re.findall(r"\b(gse_[a-z0-9_]+)\s*\(", header))
    declared -= {"gse_status"}
    assert len(declared) >= 20
    handle = lib.load()
    for name in sorted(declared):
        assert hasattr(handle, name), name
    assert declared == set(lib.PROTOTYPES), declared ^ set(lib.PROTOTYPES)
    assert handle.gse_version() == 100
    assert handle.gse_model_tag() == 0                 # the shipped library carries the bioreactor model


def test_struct_layout_matches_header(lib):
    # gse_mixture: 2 int32 + 4 + 4 + 20 + 100 floats + 100 doubles; gse_rng: 3 u64 + 2 u32
    assert ctypes.sizeof(lib.Mixture) == 8 + 4 * (4 + 4 + 20 + 100) + 8 * 100
    assert lib.Mixture.inv_cov.offset % 8 == 0
    assert ctypes.sizeof(lib.Rng) == 32


def test_loop_structs_match_the_header_as_a_c_compiler_lays_them_out(lib, tmp_path):
    """gse_loop_spec, gse_loop_tail, gse_pf_loop_spec and gse_merge_spec cross the boundary by pointer: size and every field offset of the ctypes mirrors
    against include/gse.h compiled by gcc."""
    import os
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    structs = {"gse_loop_spec": lib.LoopSpec, "gse_loop_tail": lib.LoopTail, "gse_pf_loop_spec": lib.PfLoopSpec,
               "gse_merge_spec": lib.MergeSpec, "gse_gsukf_cycles_spec": lib.GsukfCyclesSpec,
               "gse_mixture_n": lib.MixtureN}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "gse.h"', 'int main(void) {']
    for cname, cls in structs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for field, _ in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, field, cname, field))
    lines += ['return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in structs.items():
        assert int(out[cname]) == ctypes.sizeof(cls), cname
        for field, _ in cls._fields_:
            assert int(out["%s.%s" % (cname, field)]) == getattr(cls, field).offset, (cname, field)


def test_argument_errors_are_reported_without_a_gpu(lib):
    handle = lib.load()
    assert handle.gse_workspace_bytes(0, 10) < 0
    assert b"bad B" in handle.gse_last_error()
    assert handle.gse_workspace_bytes(1, 1 << 20) >= (1024 + 1025) * 8
    with pytest.raises(lib.GseError, match="null argument"):
        lib.call("gse_pf_predict_injected", None, 8, 1, 8, lib.doubles([0, 0], 2), 0.1, None, None)
    with pytest.raises(lib.GseError, match="bad ops mask"):
        lib.call("gse_pf_step", 512, 16, 16, None, None, None, 8, 1, 8, None, 0.0, None, None, None, None, None, None, None, None, None)
    with pytest.raises(lib.GseError, match="gather needs"):
        lib.call("gse_pf_step", 1, 16, None, None, None, None, 8, 1, 8, None, 0.0, None, None, None, None, None, None, None, 32, None)
    with pytest.raises(lib.GseError, match="SCAN_IN needs"):
        lib.call("gse_pf_step", 2 | 8, 16, 16, None, None, None, 8, 1, 8, None, 0.0, None, None, None, None, None, None, None, 32, None)
    with pytest.raises(lib.GseError, match="multiple of 4"):
        lib.call("gse_pf_predict_injected", 16, 7, 1, 7, lib.doubles([0, 0], 2), 0.1, 16, None)
    with pytest.raises(lib.GseError, match=r"outside \[0, 1\)"):
        lib.call("gse_pf_resample", 16, 32, 48, 64, None, 8, 1, 8, lib.doubles([1.5], 1), 0, None, 80, None)
    # the Gaussian-sum filter's entry points, the deferred resample included
    with pytest.raises(lib.GseError, match="B = 40"):
        lib.call("gse_gsukf_sigma_points", 16, 32, 48, 40, 8, None, None)
    with pytest.raises(lib.GseError, match="at most 64 Gaussians"):
        lib.call("gse_gsukf_cycle", 16, 32, 48, None, 1, 65, lib.doubles([0, 0], 2), 0.1, lib.doubles([0, 0], 2), lib.doubles([0.5], 1),
                 (lib.c_f * 11)(), None, None, 64, None, None, None)
    with pytest.raises(lib.GseError, match=r"outside \[0, 1\)"):
        lib.call("gse_gsukf_resample_begin", 16, 32, 48, 1, 100, lib.doubles([-0.1], 1), 64, None)
    with pytest.raises(lib.GseError, match="aliasing"):
        lib.call("gse_gsukf_resample_begin", 16, 16, 48, 1, 100, lib.doubles([0.5], 1), 64, None)
    with pytest.raises(lib.GseError, match="aliasing"):
        lib.call("gse_gsukf_gather_rows", 16, 32, 48, 16, 64, 1, 100, None)
    m = lib.Mixture()
    m.nd, m.nx = 2, 2
    import ctypes
    with pytest.raises(lib.GseError, match="source arrays must be given"):
        lib.call("gse_gsukf_predict_update_from", None, None, 48, 16, 32, 64, 1, 100, lib.doubles([0, 0], 2), 0.1,
                 lib.doubles([0, 0], 2), (lib.c_f * 11)(), None, None, 80, ctypes.byref(m), None, 0, None)


def test_moments_to_mean_cov():
    from gpu_se_amd.filter.particle import moments_to_mean_cov
    rng = np.random.default_rng(0)
    x = 20 + rng.standard_normal((500, 5))
    w = rng.random(500)
    p = x[0]
    d = x - p
    s2 = d.T @ (d * w[:, None])
    s0, mean, cov = moments_to_mean_cov(w.sum(), w @ d, s2[np.triu_indices(5)], p)
    np.testing.assert_allclose(mean, w @ x, rtol=1e-12)
    dist = x - (w @ x)
    np.testing.assert_allclose(cov, dist.T @ (dist * w[:, None]), rtol=1e-9, atol=1e-9)


def test_host_model_matches_oracle():
    from gpu_se_amd.model import Bioreactor
    from oracle import gse_oracle as orc
    rng = np.random.default_rng(1)
    xss = orc.steady_state()
    np.testing.assert_allclose(Bioreactor.find_SS([0.06, 0.2], [260 / 180, 640 / 24.6, 1000 / 116, 0, 0]), xss, rtol=1e-12)
    for _ in range(50):
        x = xss * (1 + 0.3 * rng.uniform(-1, 1, 5))
        x[3] = rng.uniform(-0.1, 0.3)
        u = rng.uniform(0, 0.2, 2)
        np.testing.assert_allclose(Bioreactor.homeostatic_DEs(x, u, 0.3), orc.bioreactor_f(x, u, 0.3), rtol=1e-12, atol=1e-15)
    X = xss[None, :] * (1 + 0.3 * rng.uniform(-1, 1, (20, 5)))
    U = rng.uniform(0, 0.2, (20, 2))
    rows = Bioreactor.homeostatic_DEs_rows(X, U, 0.3)
    for i in range(20):
        np.testing.assert_allclose(rows[i], Bioreactor.homeostatic_DEs(X[i], U[i], 0.3), rtol=1e-13, atol=1e-16)
    assert Bioreactor.static_outputs(xss, None) == (xss[0] * 180, xss[2] * 116)
    b = Bioreactor(xss)
    b.step(0.1, [0.06, 0.2])
    np.testing.assert_allclose(b.X, xss, atol=1e-10)


def test_mixture_host_attributes(golden, lib):
    """The drop-in MultivariateGaussianSum keeps the reference's attributes (host side) -- golden G7."""
    from gpu_se_amd import sim_base
    g = golden("constants.npz")
    sp, mp = sim_base.get_noise()
    np.testing.assert_array_equal(sp._constants, g["state_consts"])
    np.testing.assert_array_equal(mp._constants, g["meas_consts"])
    np.testing.assert_array_equal(sp._inverse_covariances, g["state_inv"])
    np.testing.assert_array_equal(mp._inverse_covariances, g["meas_inv"])
    assert (sp._Nd, sp._Nx, mp._Nd, mp._Nx) == (2, 5, 2, 2)
    sp.means += np.ones((1, 5), dtype=np.float32)          # get_parts mutates means after construction
    pod = sp._pod()
    assert pod.means[1][4] == 1.0 and pod.nd == 2 and pod.nx == 5
    assert abs(pod.chol[1][0][0] - np.sqrt(1e-3)) < 1e-8
    with pytest.raises(NotImplementedError):
        sim_base.get_parts(gpu=False)
