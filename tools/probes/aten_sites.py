"""Which source lines of the host mirror issue ATen kernels during one swap (the glue between the library's kernels):
a TorchDispatchMode counts every ATen call that launches a kernel, keyed by the innermost hairfastgan_amd frame.
python tools/probes/aten_sites.py [triples per pass, default 1]"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

import bench

VIEWS = ("view", "slice", "select", "detach", "empty", "reshape", "permute", "expand", "unsqueeze", "squeeze", "as_strided", "alias",
         "transpose", "t.default", "_unsafe_view", "unbind", "split", "chunk", "lift_fresh", "is_pinned", "_local_scalar_dense",
         "narrow", "flatten", "unflatten", "new_empty", "empty_like", "empty_strided", "resize_", "set_", "record_stream", "item")


class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.count = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(v in name for v in VIEWS):
            site = "?"
            for fr in reversed(traceback.extract_stack(limit=14)):
                if "hairfastgan_amd" in fr.filename and "_python_dispatch" not in fr.filename:
                    site = f"{os.path.relpath(fr.filename)}:{fr.lineno} {fr.line.strip()[:70]}"
                    break
            self.count[(name.replace('aten.', ''), site)] += 1
        return func(*args, **(kwargs or {}))


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
    with Sites() as m:
        run()
    torch.cuda.synchronize()
tot = sum(m.count.values())
print(f"{tot} kernel-launching ATen calls in one {'swap' if T == 1 else f'swap_batch of {T}'}")
by_op = collections.Counter()
for (op, site), n in m.count.items():
    by_op[op] += n
print("by op:", by_op.most_common(20))
for (op, site), n in m.count.most_common(70):
    print(f"{n:5d}  {op:28s} {site}")
