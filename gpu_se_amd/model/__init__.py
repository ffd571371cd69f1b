"""Host-side models: the bioreactor plant (its f / g are the functions the device kernels implement) and the
discrete linear model the MPC is built on.  Nothing here touches the GPU."""
from . import BioreactorModel as _plant
from . import LinearModel as _linear

Bioreactor = _plant.Bioreactor
LinearModel = _linear.LinearModel

__all__ = ("Bioreactor", "LinearModel")
