from gpu_se_amd.model.BioreactorModel import Bioreactor
from gpu_se_amd.model.LinearModel import LinearModel

__all__ = ['Bioreactor', 'LinearModel']
