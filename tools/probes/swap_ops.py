"""Where the launches of ONE swap come from (single HairFast.swap, batch 1): torch.profiler over one warm swap -
(a) device kernels by name and count, (b) ATen ops (elementwise glue, copies, cats) by count with the Python source line
that issued them.  python tools/probes/swap_ops.py [batch]   (batch > 1: one swap_batch pass of that many triples)"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import ProfilerActivity, profile

import bench

dev = torch.device("cuda:0")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1
g, sd = bench.build_generator(dev)
hf = bench.build_hairfast(sd, dev)
load = bench.make_triple_loader(2)
trip = [tuple(t.to(dev) for t in load(i)) for i in range(max(T, 2))]
run = (lambda: hf.swap(*trip[0])) if T == 1 else (lambda: hf.swap_batch(trip[:T]))
with torch.inference_mode():
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        run()
        torch.cuda.synchronize()
ev = prof.events()
kern = collections.Counter()
ktime = collections.Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CUDA:
        kern[e.name[:70]] += 1
        ktime[e.name[:70]] += e.device_time if hasattr(e, "device_time") else e.cuda_time
print(f"== device kernels: {sum(kern.values())} launches, {sum(ktime.values()) / 1e3:.2f} ms of GPU time")
for k, n in kern.most_common(45):
    print(f"{n:6d}  {ktime[k] / 1e3:8.3f} ms  {k}")
ops = collections.Counter()
where = collections.defaultdict(collections.Counter)
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith("aten::"):
        continue
    if e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::"):
        continue  # top-level ATen calls only
    ops[e.name] += 1
    site = next((s for s in (e.stack or []) if "hairfastgan_amd" in s), (e.stack or ["?"])[0] if e.stack else "?")
    where[e.name][site.replace(root + "/", "")[:110]] += 1
print("== top-level ATen ops by count")
for k, n in ops.most_common(25):
    print(f"{n:6d}  {k}")
    for site, m in where[k].most_common(8):
        print(f"          {m:5d}  {site}")
