"""Gaussian-sum noise models evaluated and sampled on the MI355X (device counterparts of the reference's
``gaussian_sum_dist`` package; the C ABI behind them is ``gse_mixture_pdf_*`` / ``gse_mixture_draw``)."""
from . import DeterministicGaussianSum as _deterministic
from . import MultivariateGaussianSum as _multivariate

MultivariateGaussianSum = _multivariate.MultivariateGaussianSum
DeterministicGaussianSum = _deterministic.DeterministicGaussianSum

__all__ = ("MultivariateGaussianSum", "DeterministicGaussianSum")
