"""Which C-ABI calls one batched swap makes, by shape: wraps every _marshal entry point that launches a kernel, brackets
each call with events and prints the shape classes by total time.  (Event bracketing serialises nothing - one stream -
but adds ~2 x 5 us of host work per call; the times are for ranking, bench.py's are the measurement.)"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench  # noqa: E402
from hairfastgan_amd import _marshal as M  # noqa: E402

NAMES = [n for n in dir(M) if callable(getattr(M, n)) and not n.startswith("_") and n not in
         ("check", "conv2d_f16_supported", "conv1x1_f16_supported", "modconv3x3_small_supported", "modconv3x3_f16_supported",
          "modconv3x3_up_f16_supported", "SplitActivation")]
REC = []


def shape_of(a):
    if torch.is_tensor(a):
        return tuple(a.shape)
    if isinstance(a, M.SplitActivation):
        return ("split",) + tuple(a.shape)
    return None


def wrap(name, fn):
    def inner(*a, **k):
        if not ACTIVE[0]:
            return fn(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*a, **k)
        e1.record()
        shapes = tuple(s for s in (shape_of(x) for x in a[2:6]) if s is not None)
        extra = tuple((kk, k[kk]) for kk in ("groups", "upsample") if kk in k and not torch.is_tensor(k[kk]))
        ints = tuple(x for x in a[2:9] if isinstance(x, int))
        REC.append((name, shapes[:2], ints, extra, e0, e1))
        return out
    return inner


ACTIVE = [False]
import types  # noqa: E402

for n in NAMES:
    f = getattr(M, n)
    if isinstance(f, types.FunctionType):
        setattr(M, n, wrap(n, f))

dev = torch.device("cuda:0")
_g, sd = bench.build_generator(dev)
hf = bench.build_hairfast(sd, dev)
load = bench.make_triple_loader(2)
nb = int(os.environ.get("SWAP_BATCH", "8"))
trip = [tuple(t.to(dev) for t in load(i)) for i in range(nb)]
with torch.inference_mode():
    hf.swap(*trip[0])
    hf.swap_batch(trip)
    torch.cuda.synchronize()
    ACTIVE[0] = True
    hf.swap_batch(trip)
    torch.cuda.synchronize()
    ACTIVE[0] = False
tot = collections.defaultdict(lambda: [0, 0.0])
for name, shapes, ints, extra, e0, e1 in REC:
    key = (name, shapes, ints, extra)
    tot[key][0] += 1
    tot[key][1] += e0.elapsed_time(e1)
allms = sum(v[1] for v in tot.values())
print(f"{len(REC)} calls, {allms:.1f} ms inside marshal calls for {nb} triples")
byname = collections.defaultdict(lambda: [0, 0.0])
for (name, *_), v in tot.items():
    byname[name][0] += v[0]
    byname[name][1] += v[1]
for name, v in sorted(byname.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {name:32s} {v[0]:5d} calls {v[1]:8.2f} ms")
print("top shape classes:")
for key, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get("TOP", "140"))]:
    print(f"  {v[1]:7.2f} ms {v[0]:4d}x {key[0]} {key[1]} {key[2]} {key[3]}")
