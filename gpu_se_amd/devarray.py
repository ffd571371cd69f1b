"""Device arrays handed out by the drop-in classes.

The reference returns CuPy arrays and its callers use ``.get()`` to come back to the host
(sim_base.py:281-284, filter/particle.py:320).  Here device memory is a torch (ROCm) tensor; DevArray is a
zero-cost torch.Tensor subclass that adds ``.get()`` and NumPy conversion, nothing else.
"""
import numpy
import torch


class DevArray(torch.Tensor):
    @staticmethod
    def wrap(t):
        return t.as_subclass(DevArray)

    def get(self):
        """Synchronising copy to a NumPy array (CuPy's ndarray.get)."""
        return self.detach().as_subclass(torch.Tensor).cpu().numpy()

    def __array__(self, dtype=None, copy=None):
        a = self.get()
        return a.astype(dtype) if dtype is not None else a


def device():
    if not torch.cuda.is_available():
        raise RuntimeError("gpu_se_amd needs a HIP device (torch.cuda.is_available() is False); "
                           "there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


try:                                    # raw hipStream_t of torch's current stream without building a Stream object
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:                  # pragma: no cover
    _raw_stream = None


def stream_ptr(index=None):
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device() if index is None else index)
    return torch.cuda.current_stream().cuda_stream


def as_device(x, dtype):
    """Host array-like or device tensor -> contiguous device tensor of dtype."""
    if isinstance(x, torch.Tensor):
        return x.as_subclass(torch.Tensor).to(device=device(), dtype=dtype).contiguous()
    return torch.as_tensor(numpy.ascontiguousarray(numpy.asarray(x)), device=device()).to(dtype).contiguous()
