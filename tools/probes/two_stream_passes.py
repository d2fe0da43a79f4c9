"""Two batched passes in flight on two HIP streams against the same passes one after the other: does the latency-bound quarter of a
pass (glue, small launches, tails of under-filled grids) hide under the other pass's matrix kernels?  Resident device inputs;
64 triples per measurement.  Results are compared bit for bit with the sequential passes."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
g, sd = bench.build_generator(dev)
hf = bench.build_hairfast(sd, dev)
load = bench.make_triple_loader()
N = int(os.environ.get("PROBE_N", "64"))
triples = [tuple(t.to(dev) for t in load(i)) for i in range(N)]


def run(pass_size, n_streams):
    streams = [torch.cuda.Stream(dev) for _ in range(n_streams)] if n_streams > 1 else [torch.cuda.current_stream()]
    outs = [None] * N
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.inference_mode():
        for k, j in enumerate(range(0, N, pass_size)):
            s = streams[k % len(streams)]
            if n_streams > 1:
                s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                res = hf.swap_batch(triples[j:j + pass_size])
                for i, im in enumerate(res):
                    outs[j + i] = im
    torch.cuda.synchronize()
    return time.perf_counter() - t0, outs


with torch.inference_mode():
    hf.swap_batch(triples[:32])
    hf.swap_batch(triples[:16])
    hf.swap_batch(triples[:8])
ref = None
for pass_size, n_streams in [(32, 1), (16, 1), (16, 2), (32, 2), (8, 2), (8, 4), (16, 1), (16, 2)]:
    try:
        dt, outs = run(pass_size, n_streams)
    except Exception as e:  # noqa: BLE001
        print(f"pass {pass_size:2d} x {n_streams} stream(s): FAILED {type(e).__name__}: {str(e)[:200]}", flush=True)
        continue
    if ref is None:
        ref = [o.clone() for o in outs]
    same = all(torch.equal(a, b) for a, b in zip(outs, ref))
    print(f"pass {pass_size:2d} x {n_streams} stream(s): {dt * 1e3:8.1f} ms for {N} triples = {N / dt:6.1f} triples/s, "
          f"equal to the first run: {same}, peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
