"""Interface of a plant the linearisation and the closed loop can drive (reference model/NonlinearModel.py): an object
with a state ``X`` and the three methods below.  ``LinearModel.create_LinearModel`` needs nothing else."""
import abc


class NonlinearModel(abc.ABC):
    """A plant with state ``X``: its derivative, one integration step, and its outputs."""

    @abc.abstractmethod
    def DEs(self, inputs):
        """dX/dt at the current state for the given inputs."""

    @abc.abstractmethod
    def step(self, dt, inputs):
        """Advance the state by ``dt`` under constant ``inputs``."""

    @abc.abstractmethod
    def outputs(self, inputs):
        """All outputs (states and calculated quantities) at the current state."""
