"""Device / stream plumbing shared by the public ops."""
import torch

from . import _lib


def lib():
    return _lib.load()


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "hairfastgan_amd ops run on the MI355X only (got a CPU tensor); there is no CPU fallback - "
                "the CPU restatement lives in oracle/ and is test infrastructure")


def stream():
    """Raw hipStream_t of torch's current stream (kernels are enqueued asynchronously on it,
    like the reference's at::cuda::getCurrentCUDAStream(), upfirdn2d_kernel.cu:213-215)."""
    return torch.cuda.current_stream().cuda_stream
