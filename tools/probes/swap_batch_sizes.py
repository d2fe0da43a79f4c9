"""Throughput of HairFast.swap_batch against the number of triples per pass (resident device inputs)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench

dev = torch.device("cuda:0")
g, sd = bench.build_generator(dev)
hf = bench.build_hairfast(sd, dev)
load = bench.make_triple_loader()
for T in [int(t) for t in os.environ.get("PROBE_T", "1,4,8,12,16").split(",")]:
    triples = [tuple(t.to(dev) for t in load(i)) for i in range(T)]
    with torch.inference_mode():
        hf.swap_batch(triples)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = max(1, 16 // T)
        for _ in range(n):
            hf.swap_batch(triples)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (n * T)
    print(f"T={T:2d}: {dt * 1e3:7.2f} ms per triple  {1 / dt:6.1f} triples/s  peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
