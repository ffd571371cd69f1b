"""State estimators whose predict / update / resample cycle runs as HIP kernels on the MI355X: the device halves of the
reference's ``filter`` package plus the batched form used for Monte-Carlo rollouts."""
from .gs_ukf import BatchedGaussianSumUnscentedKalmanFilter, ParallelGaussianSumUnscentedKalmanFilter
from .particle import BatchedParticleFilter, ParallelParticleFilter

__all__ = ("ParallelParticleFilter", "BatchedParticleFilter", "ParallelGaussianSumUnscentedKalmanFilter",
           "BatchedGaussianSumUnscentedKalmanFilter")
