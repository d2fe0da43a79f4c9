"""hipGraph capture of hot-path calls (launch-bound regime: batch 1-3).

A generator forward at batch 1 is ~110 kernel launches in ~2.5 ms and an e4e forward
~250 launches: the host (Python + ctypes) issues a launch every 5-10 us, so at small
batch the GPU waits for the CPU.  `GraphRunner` records one call of `fn` on static input
buffers into a HIP graph (through torch.cuda.CUDAGraph: every kernel of this library is
enqueued on torch's current stream, so stream capture sees all of them, including the
`normal_()` noise draws, which torch makes graph-safe) and replays it with one launch.

The returned tensors are the graph's static output buffers: they are overwritten by the
next replay, exactly like the outputs of torch.cuda.make_graphed_callables.
"""
import torch


def _flatten(out):
    if torch.is_tensor(out) or out is None:
        return [out]
    flat = []
    for o in out:
        flat += _flatten(o)
    return flat


class GraphRunner:
    def __init__(self, fn, *example_inputs, warmup=2):
        if not example_inputs or not all(torch.is_tensor(a) and a.is_cuda for a in example_inputs):
            raise ValueError("GraphRunner needs GPU tensor arguments (shapes are frozen into the graph)")
        self.fn = fn
        self.static_in = [a.detach().clone() for a in example_inputs]
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side), torch.inference_mode():
            for _ in range(warmup):  # derive cached weights / folded BN outside the capture
                fn(*self.static_in)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.inference_mode(), torch.cuda.graph(self.graph):
            self.static_out = fn(*self.static_in)

    def __call__(self, *inputs):
        if len(inputs) != len(self.static_in):
            raise ValueError("argument count differs from the captured call")
        # inference mode like the capture: the graph's RNG bookkeeping (philox offset of the draws recorded inside) is an
        # inference tensor that replay() updates in place
        with torch.inference_mode():
            for dst, src in zip(self.static_in, inputs):
                if dst.shape != src.shape:
                    raise ValueError(f"shape {tuple(src.shape)} differs from the captured {tuple(dst.shape)}")
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src)
            self.graph.replay()
        return self.static_out
