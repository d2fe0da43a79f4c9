"""GPU time by CALL SITE of one batched swap pass: every ATen call that launches a kernel (TorchDispatchMode) and every
hairfastgan_amd._marshal function is bracketed by HIP events and keyed by (operation, innermost hairfastgan_amd source line,
first tensor shape) - where the glue between the matrix-core kernels is, in milliseconds.
python tools/probes/time_sites.py [triples per pass, default 32] [rows to print, default 70]"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

import bench
from hairfastgan_amd import _marshal as M

VIEWS = ("view", "slice", "select", "detach", "empty", "reshape", "permute", "expand", "unsqueeze", "squeeze", "as_strided", "alias",
         "transpose", "t.default", "_unsafe_view", "unbind", "split", "chunk", "lift_fresh", "is_pinned", "_local_scalar_dense",
         "narrow", "flatten", "unflatten", "new_empty", "empty_like", "empty_strided", "resize_", "set_", "record_stream", "item")
EVENTS = []  # (kind, name, site, shape, e0, e1)
ON = [False]
DEPTH = [0]


def site_of():
    for fr in reversed(traceback.extract_stack(limit=18)):
        if "hairfastgan_amd" in fr.filename and "_marshal.py" not in fr.filename and "_python_dispatch" not in fr.filename:
            return f"{os.path.relpath(fr.filename)}:{fr.lineno} {fr.line.strip()[:64]}"
    return "?"


def shape_of(args):
    for a in args:
        if torch.is_tensor(a):
            return tuple(a.shape)
        if isinstance(a, M.SplitActivation):
            return ("split",) + tuple(a.shape)
        if isinstance(a, (list, tuple)) and a and torch.is_tensor(a[0]):
            return tuple(a[0].shape)
    return ()


def bracket(kind, name, fn, args, kwargs):
    if not ON[0] or DEPTH[0] > 0:
        return fn(*args, **kwargs)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    DEPTH[0] += 1
    try:
        e0.record()
        r = fn(*args, **kwargs)
        e1.record()
    finally:
        DEPTH[0] -= 1
    EVENTS.append((kind, name, site_of(), shape_of(args), e0, e1))
    return r


class Sites(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(v in name for v in VIEWS):
            return func(*args, **(kwargs or {}))
        return bracket("aten", name.replace("aten.", ""), func, args, kwargs or {})


for n in dir(M):
    f = getattr(M, n)
    if callable(f) and not n.startswith("_") and getattr(f, "__module__", "") == M.__name__ and not isinstance(f, type):
        setattr(M, n, (lambda name, fn: (lambda *a, **k: bracket("hf", name, fn, a, k)))(n, f))

dev = torch.device("cuda:0")
GEN = len(sys.argv) > 1 and sys.argv[1] == "gen"  # `time_sites.py gen [rows]`: one generator 0->8 forward at batch 8 instead of a swap
T = 8 if GEN else (int(sys.argv[1]) if len(sys.argv) > 1 else 32)
ROWS = int(sys.argv[2]) if len(sys.argv) > 2 else 70
g, sd = bench.build_generator(dev)
if GEN:
    lat = torch.randn(T, 18, 512, device=dev)
    run = lambda: g([lat], input_is_latent=True)  # noqa: E731
else:
    hf = bench.build_hairfast(sd, dev)
    load = bench.make_triple_loader(4)
    trip = [tuple(t.to(dev) for t in load(i)) for i in range(max(T, 2))]
    run = (lambda: hf.swap(*trip[0])) if T == 1 else (lambda: hf.swap_batch(trip[:T]))
with torch.inference_mode():
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    ON[0] = True
    with Sites():
        run()
    ON[0] = False
    torch.cuda.synchronize()
acc = collections.defaultdict(lambda: [0.0, 0])
for kind, name, site, shape, e0, e1 in EVENTS:
    a = acc[(kind, name, site, shape)]
    a[0] += e0.elapsed_time(e1)
    a[1] += 1
total = sum(v[0] for v in acc.values())
by_kind = collections.Counter()
by_name = collections.Counter()
for (kind, name, site, shape), (ms, n) in acc.items():
    by_kind[kind] += ms
    by_name[(kind, name)] += ms
print(f"{len(EVENTS)} bracketed calls, {total:.1f} ms of bracketed GPU time in one "
      f"{'generator forward at batch 8' if GEN else ('swap' if T == 1 else f'swap_batch of {T}')} (event overhead included)")
print("by kind:", {k: round(v, 1) for k, v in by_kind.items()})
print("by operation:", [(k[1], round(v, 2)) for k, v in by_name.most_common(40)])
for (kind, name, site, shape), (ms, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:ROWS]:
    print(f"{ms:8.3f} ms {n:4d}x {kind:4s} {name:26s} {str(shape):28s} {site}")
