from gpu_se_amd.filter.particle import BatchedParticleFilter, ParallelParticleFilter
from gpu_se_amd.filter.gs_ukf import ParallelGaussianSumUnscentedKalmanFilter

__all__ = ['ParallelParticleFilter', 'BatchedParticleFilter', 'ParallelGaussianSumUnscentedKalmanFilter']
