from gpu_se_amd.gaussian_sum_dist.MultivariateGaussianSum import MultivariateGaussianSum
from gpu_se_amd.gaussian_sum_dist.DeterministicGaussianSum import DeterministicGaussianSum

__all__ = ['MultivariateGaussianSum', 'DeterministicGaussianSum']
