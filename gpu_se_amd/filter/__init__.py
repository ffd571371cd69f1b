from gpu_se_amd.filter.particle import ParallelParticleFilter
from gpu_se_amd.filter.gs_ukf import ParallelGaussianSumUnscentedKalmanFilter

__all__ = ['ParallelParticleFilter', 'ParallelGaussianSumUnscentedKalmanFilter']
