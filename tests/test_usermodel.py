"""User models, CPU side: translation of restricted Python to HIP, and that the particle-filter
library builds for a translated model (hipcc cross-compiles without a GPU) and carries the model's tag."""
import ctypes
import os

import numpy as np
import pytest

from gpu_se_amd import usermodel
import usermodels as um


def test_translation_of_the_reference_style_pair():
    nx, nu, text = usermodel.translate(um.two_state_f, um.two_state_g)
    assert (nx, nu) == (2, 1)
    assert "user_increment" in text and "user_output" in text and "fp contract(off)" in text
    assert "dx_[2] = 0.0;" in text and "dx_[4] = 0.0;" in text              # unused states stay where they are
    nx, nu, text = usermodel.translate(um.three_state_f, um.three_state_g)
    assert (nx, nu) == (3, 2)
    for piece in ("fmax(", "fabs(", "exp(", "sqrt(", "fmin(", "?"):
        assert piece in text


def test_unsupported_python_is_refused():
    with pytest.raises(usermodel.ModelError, match="outside the supported form"):
        usermodel.translate(um.loop_f, um.two_state_g)
    with pytest.raises(usermodel.ModelError, match="exactly 2 outputs"):
        usermodel.translate(um.two_state_f, _one_output)
    with pytest.raises(usermodel.ModelError):
        usermodel.translate(lambda x, u, dt: (x[0] * dt,), um.two_state_g)   # not a plain function definition


def _one_output(x, u):
    return x[0] + u[0],


def test_kernels_compile_at_run_time_for_a_user_model():
    """The default backend: hiprtc in the process, one kernel instance at a time.  Compilation needs no device (loading the
    code object does: tests/test_gpu_usermodel.py)."""
    model = usermodel.compile_model(um.two_state_f, um.two_state_g)
    assert isinstance(model, usermodel.RuntimeModel) and (model.nx, model.nu) == (2, 1) and model.kernels == 0
    for name in ("gse::pf_step_kernel<true, true, false, true, 2, 3>",        # the single-launch cycle
                 "gse::pf_step_kernel<false, true, false, false, 0, 0>",     # predict alone
                 "gse::model_f_kernel", "gse::model_g_kernel"):
        assert model.compile_check(name) > 10000
    with pytest.raises(Exception, match="failed"):
        model.compile_check("gse::no_such_kernel")
    assert usermodel.compile_model(um.two_state_f, um.two_state_g) is model          # cached
    three = usermodel.compile_model(um.three_state_f, um.three_state_g)
    assert three.nx == 3
    # the GS-UKF kernels (a translation unit of their own) for a model with 3 states: 7 sigma points, 3 x 3 covariances
    for name in ("gse::gsukf_cycle_kernel<false>", "gse::gsukf_predict_kernel<true, 64>", "gse::gsukf_update_kernel<64>",
                 "gse::gsukf_predict_update_lanes_kernel<false, 1>", "gse::gsukf_predict_update_lanes_kernel<false, 0>",
                 "gse::gsukf_sigma_kernel<0>", "gse::gsukf_moments_kernel<0>",
                 "gse::gather_kernel<gse::GsukfPayload>"):
        assert three.compile_check(name) > 5000


def test_library_builds_for_a_user_model():
    """The other backend: the whole particle-filter library rebuilt by the hipcc binary (cached on disk)."""
    model = usermodel.compile_model(um.two_state_f, um.two_state_g, backend="hipcc")
    assert os.path.exists(model.path) and (model.nx, model.nu) == (2, 1)
    assert int(model.lib.gse_model_tag()) == model.tag != 0
    for name in ("gse_pf_step", "gse_pf_resample_begin", "gse_pf_moments", "gse_model_f", "gse_model_g_u", "gse_last_error",
                 "gse_gsukf_cycle", "gse_gsukf_predict_update", "gse_gsukf_moments"):
        assert hasattr(model.lib, name), name
    # argument checking works in the model's library too, without a GPU
    assert model.lib.gse_pf_step(512, 16, 16, None, None, None, 8, 1, 8, None, 0.0, None, None, None, None, None, None,
                                 None, None, None) < 0
    assert b"bad ops mask" in model.lib.gse_last_error()
    assert usermodel.compile_model(um.two_state_f, um.two_state_g, backend="hipcc") is model          # cached


def test_models_beyond_the_fast_bounds_get_the_generic_kernels():
    """Round 5: up to 8 states / 4 inputs / 4 outputs through the dimension-generic kernels (csrc/gse_generic_kernels.h), compiled
    by hiprtc with the model's dimensions as constants -- no device needed for the compilation itself."""
    assert usermodel.model_dimensions(um.six_state_f, um.six_state_g) == (6, 3, 3)
    assert usermodel.needs_generic_kernels(um.six_state_f, um.six_state_g)
    assert not usermodel.needs_generic_kernels(um.two_state_f, um.two_state_g)
    assert usermodel.needs_generic_kernels(um.two_state_f, _one_output)              # one output: not the fast kernels' two
    nx, nu, ny, text = usermodel.translate_generic(um.six_state_f, um.six_state_g)
    assert (nx, nu, ny) == (6, 3, 3)
    for piece in ("#define GSE_G_NX 6", "#define GSE_G_NU 3", "#define GSE_G_NY 3", "generic_increment", "generic_output",
                  "fp contract(off)", "fmin("):
        assert piece in text
    model = usermodel.compile_generic_model(um.six_state_f, um.six_state_g)
    assert usermodel.compile_generic_model(um.six_state_f, um.six_state_g) is model
    for name in ("generic_pf_predict_kernel<false>", "generic_pf_update_kernel", "generic_gsukf_predict_kernel<true>",
                 "generic_gsukf_update_kernel", "generic_moments_kernel"):
        assert model.compile_check(name) > 10000
    with pytest.raises(usermodel.ModelError, match="outside the supported form|beyond"):
        usermodel.needs_generic_kernels(_nine_state_f, um.six_state_g)


def _nine_state_f(x, u, dt):
    return x[0] * dt, x[1] * dt, x[2] * dt, x[3] * dt, x[4] * dt, x[5] * dt, x[6] * dt, x[7] * dt, x[8] * dt


def test_wide_mixture_host_side():
    """A MultivariateGaussianSum of 6 dimensions keeps the reference's host attributes and packs the generic kernels' struct."""
    from gpu_se_amd import gaussian_sum_dist as gsd
    cov = np.array([np.diag([1e-2, 2e-2, 1e-2, 3e-2, 1e-2, 2e-2]), np.diag([1e-2] * 6) + 1e-3])
    m = gsd.MultivariateGaussianSum(means=np.zeros((2, 6)), covariances=cov, weights=np.array([0.7, 0.9]))
    assert (m._Nd, m._Nx, m._wide) == (2, 6, True)
    pod = m._pod_n()
    assert pod.nd == 2 and pod.n == 6 and abs(pod.cumw[0] - 0.7 / 1.6) < 1e-6 and abs(pod.cumw[1] - 1.0) < 1e-6
    np.testing.assert_allclose(np.array(pod.chol[1])[:6, :6], np.linalg.cholesky(cov[1]), rtol=1e-6)
    np.testing.assert_allclose(np.array(pod.inv[0])[:6, :6], np.linalg.inv(cov[0]), rtol=1e-12)
    np.testing.assert_allclose(pod.cw[1], (2 * np.pi) ** -3 / np.sqrt(np.linalg.det(cov[1])) * 0.9, rtol=1e-6)
    with pytest.raises(ValueError, match="no fast-kernel struct"):
        m._pod()
    with pytest.raises(ValueError):
        gsd.MultivariateGaussianSum(means=np.zeros((1, 9)), covariances=np.array([np.eye(9)]), weights=np.array([1.0]))
