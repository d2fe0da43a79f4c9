"""Host-side cost of ONE HairFast.swap (eager): cProfile of the Python between the launches - a single swap enqueues ~2100
launches, and at ~15 us of host time each the host, not the GPU, is what an eager swap waits for (the hipGraph replay of the
same swap is 3-5 ms faster)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
g, sd = bench.build_generator(dev)
hf = bench.build_hairfast(sd, dev)
load = bench.make_triple_loader()
trip = tuple(t.to(dev) for t in load(0))
with torch.inference_mode():
    for _ in range(3):
        hf.swap(*trip)
    torch.cuda.synchronize()
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        hf.swap(*trip)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"eager swap: host enqueue {t_enq / n * 1e3:.2f} ms, wall incl. GPU drain {t_all / n * 1e3:.2f} ms per swap")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        hf.swap(*trip)
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumulative").print_stats(45)
