from gpu_se_amd.model.BioreactorModel import Bioreactor

__all__ = ['Bioreactor']
