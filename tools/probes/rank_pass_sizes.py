"""One rank's share of BASELINE.json configs[3] on 8 GPUs (256 triples / 8 = 32 per rank) at different pass sizes, on ONE GPU
with RCCL initialised (HF_FORCE_DIST=1, world 1): wall from the first H2D to the last gather as `parallel.swap_many`
reports it - compute (incl. the exposed copy-in of the first pass), the exposed tail of the chunked all-gather - to pick the
pass size per rank count (round-4 verdict item 7).  PROBE_TRIPLES / PROBE_BATCHES override 32 and 32,16,8."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HF_FORCE_DIST", "1")
for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29655"), ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
    os.environ.setdefault(k, v)
import bench  # noqa: E402
from hairfastgan_amd import parallel  # noqa: E402

rank, world, local = parallel.init_from_env()
dev = torch.device("cuda", local)
g, sd = bench.build_generator(dev)
hf = bench.build_hairfast(sd, dev)
load = bench.make_triple_loader()
N = int(os.environ.get("PROBE_TRIPLES", "32"))
with torch.inference_mode():
    hf.swap(*[t.to(dev) for t in load(0)])
    for B in [int(b) for b in os.environ.get("PROBE_BATCHES", "32,16,8").split(",")]:
        hf.swap_batch([tuple(t.to(dev) for t in load(i)) for i in range(B)])  # plans / weight splits of this pass size
        torch.cuda.synchronize()
        best = None
        for rep in range(2):
            st = {}
            t0 = time.perf_counter()
            parallel.swap_many(lambda a, b, c: hf.swap(a, b, c), N, load, device=dev, batch=B, swap_batch_fn=hf.swap_batch, stats=st)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            if best is None or wall < best[0]:
                best = (wall, st["compute_s"], st["gather_tail_s"])
        # the same passes on resident inputs, no gather: what the copy-in and the gather add
        trip = [tuple(t.to(dev) for t in load(i)) for i in range(B)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(N // B):
            hf.swap_batch(trip)
        torch.cuda.synchronize()
        resident = time.perf_counter() - t0
        del trip
        print(f"{N} triples, {B:2d} per pass ({N // B} passes): wall {best[0] * 1e3:7.1f} ms = {N / best[0]:5.1f} triples/s "
              f"(compute incl. first copy-in {best[1] * 1e3:7.1f}, gather tail {best[2] * 1e3:5.1f}); resident passes only "
              f"{resident * 1e3:7.1f} ms -> exposed copy-in + gather {100 * (best[0] - resident) / best[0]:4.1f} % of the wall", flush=True)
import torch.distributed as dist  # noqa: E402

dist.barrier()
dist.destroy_process_group()
