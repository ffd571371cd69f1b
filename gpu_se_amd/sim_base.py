"""Entry points of the reference's sim_base.py for the estimator path: get_parts, get_noise, get_random_io.

What is NOT here (out of scope): the linear model and the OSQP-based MPC that the
reference's get_parts also builds -- they are returned as None.  The closed-loop Simulation driver is a later
row of the coverage table.
"""
import numpy

from gpu_se_amd import filter as _filter
from gpu_se_amd import gaussian_sum_dist, model


def get_parts(dt_control=1, N_particles=2 * 15, gpu=True, pf=True):
    """Returns the parts needed for a closed-loop simulation (sim_base.py:10-114).

    Returns
    -------
    bioreactor : model.Bioreactor   the nonlinear plant at its steady state
    lin_model : None                (linearisation for the MPC: out of scope here)
    K : None                        (MPC controller: out of scope here)
    pf : filter.ParallelParticleFilter or filter.ParallelGaussianSumUnscentedKalmanFilter
    """
    if not gpu:
        raise NotImplementedError("gpu_se_amd provides the device classes only (gpu=True); the NumPy classes of "
                                  "the reference are restated under oracle/ as test infrastructure")
    _ = dt_control
    bioreactor = model.Bioreactor(
        X0=model.Bioreactor.find_SS(
            numpy.array([0.06, 0.2]),
            #            Ng,         Nx,      Nfa, Ne, Nh
            numpy.array([260 / 180, 640 / 24.6, 1000 / 116, 0, 0])),
        high_N=False)

    my_filter = _filter.ParallelParticleFilter if pf else _filter.ParallelGaussianSumUnscentedKalmanFilter
    state_pdf, measurement_pdf = get_noise()
    x0, _ = get_noise()
    x0.means += numpy.asarray(bioreactor.X[numpy.newaxis, :], dtype=numpy.float32)     # prior centred on X_ss (:104)
    estimator = my_filter(
        f=bioreactor.homeostatic_DEs,
        g=bioreactor.static_outputs,
        N_particles=N_particles,
        x0=x0,
        state_pdf=state_pdf,
        measurement_pdf=measurement_pdf)
    return bioreactor, None, None, estimator


def get_noise(lib=None, deterministic=False):
    """Returns state and measurement noise (sim_base.py:117-161).  ``lib`` is accepted for signature
    compatibility; the distributions always evaluate on the HIP device."""
    distribution = (gaussian_sum_dist.DeterministicGaussianSum if deterministic
                    else gaussian_sum_dist.MultivariateGaussianSum)
    state_pdf = distribution(
        means=numpy.zeros(shape=(2, 5)),
        covariances=numpy.array([numpy.diag([1e-4, 1e-7, 1e-3, 1e-3, 1e-7]),
                                 numpy.diag([1e-3, 1e-6, 1e-2, 1e-2, 1e-6])]),
        weights=numpy.array([0.75, 0.25]),
        library=lib)
    measurement_pdf = distribution(
        means=numpy.array([[1e-1, 0], [0, -1e-1]]),
        covariances=numpy.array([[[6e-2, 0], [0, 8e-2]],
                                 [[500, 100], [100, 700]]]),
        weights=numpy.array([0.85, 0.15]),
        library=lib)
    return state_pdf, measurement_pdf


def get_random_io(rng=None):
    """Random system input and output for benchmark runs (sim_base.py:188-204).  NOTE: the
    reference's y is in g/L while g() returns mg/L, so every likelihood underflows with it; benchmarks here
    use get_realistic_io instead."""
    rnd = numpy.random if rng is None else rng
    u = numpy.array([rnd.uniform(low=0, high=0.1), rnd.uniform(low=0, high=0.2)])
    y = numpy.array([rnd.uniform(low=0.25, high=0.3), rnd.uniform(low=0.8, high=0.9)])
    return u, y


def get_realistic_io(rng, x_ref):
    """u as get_random_io; z = g(x_ref) + a host draw of the measurement noise (for benchmarks / tests)."""
    u = numpy.array([rng.uniform(0, 0.1), rng.uniform(0, 0.2)])
    y = numpy.array(model.Bioreactor.static_outputs(x_ref, u), dtype=numpy.float64)
    if rng.uniform() < 0.85:
        y = y + numpy.array([0.1, 0.0]) + rng.standard_normal(2) * numpy.sqrt([6e-2, 8e-2])
    else:
        y = y + numpy.array([0.0, -0.1]) + numpy.linalg.cholesky(numpy.array([[500., 100.], [100., 700.]])) @ rng.standard_normal(2)
    return u, y
