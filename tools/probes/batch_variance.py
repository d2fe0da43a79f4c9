"""Batch-invariant mode bisect: which C-ABI call gives sample i different bits at batch B and batch 2B?
Wraps every function of hairfastgan_amd._marshal that returns tensors, runs (a) the e4e encoder on 3 images and on those 3
images twice (6), (b) the shape adaptor on 2 and 4 pairs, and reports the first calls whose per-sample outputs differ."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from hairfastgan_amd import _marshal as M
from hairfastgan_amd import _runtime
from oracle import cases as C

dev = torch.device("cuda:0")
_runtime.set_batch_invariant(True)
g, sd = bench.build_generator(dev)
hf = bench.build_hairfast(sd, dev)
LOG = None


def wrap(name, fn):
    def inner(*a, **k):
        r = fn(*a, **k)
        if LOG is not None:
            outs = r if isinstance(r, (tuple, list)) else (r,)
            ts = [t for t in outs if torch.is_tensor(t)]
            ts += [t.hi for t in outs if isinstance(t, M.SplitActivation)]
            shapes = [tuple(x.shape) for x in a if torch.is_tensor(x)][:2]
            LOG.append((name, shapes, [t.clone() for t in ts]))
        return r
    return inner


for n in dir(M):
    f = getattr(M, n)
    if callable(f) and not n.startswith("_") and getattr(f, "__module__", "") == M.__name__ and not isinstance(f, type):
        setattr(M, n, wrap(n, f))


def record(fn):
    global LOG
    LOG = []
    with torch.inference_mode():
        out = fn()
    torch.cuda.synchronize()
    log, LOG = LOG, None
    return out, log


def compare(tag, log_a, log_b, rows_a, rows_b, mult):
    """rows_a[i] of run A <-> rows_b[i] of run B on the leading dimension (scaled by `mult` when a call folds samples)."""
    print(f"== {tag}: {len(log_a)} / {len(log_b)} calls")
    if len(log_a) != len(log_b):
        print("   call sequences differ in length:", [x[0] for x in log_a][:40], [x[0] for x in log_b][:40])
    shown = 0
    for k, ((na, sa, ta), (nb, sb, tb)) in enumerate(zip(log_a, log_b)):
        if na != nb:
            print(f"   call {k}: {na} vs {nb} - different dispatch"); shown += 1
            continue
        for j, (x, y) in enumerate(zip(ta, tb)):
            for lead in (0, 1):
                if (x.dim() > lead and y.dim() > lead and x.shape[lead] * 2 == y.shape[lead] and x.shape[:lead] == y.shape[:lead]
                        and x.shape[lead + 1:] == y.shape[lead + 1:]):
                    xa = x.movedim(lead, 0); yb = y.movedim(lead, 0)
                    n = xa.shape[0]
                    d1 = (xa.float() - yb[:n].float()).abs().amax(dim=tuple(range(1, xa.dim()))) if xa.dim() > 1 else (xa.float() - yb[:n].float()).abs()
                    d2 = (xa.float() - yb[n:].float()).abs().amax(dim=tuple(range(1, xa.dim()))) if xa.dim() > 1 else (xa.float() - yb[n:].float()).abs()
                    if float(d1.max()) > 0 or float(d2.max()) > 0:
                        print(f"   call {k}: {na} in {sa} out{j} {tuple(x.shape)} vs {tuple(y.shape)} lead {lead}: first copy {d1.tolist()} second copy {d2.tolist()}")
                        shown += 1
                    break
        if shown >= 12:
            break


x3 = torch.randn(3, 3, 256, 256, device=dev) * 0.5
from hairfastgan_amd.encoders import get_latents

with torch.inference_mode():  # warm-up: plan-time calls (weight preparation) stay out of the recorded sequences
    get_latents(hf.embed.e4e, x3)
    get_latents(hf.embed.e4e, torch.cat([x3, x3], 0))
_, la = record(lambda: get_latents(hf.embed.e4e, x3))
_, lb = record(lambda: get_latents(hf.embed.e4e, torch.cat([x3, x3], 0)))
compare("e4e batch 3 vs 6 (second copy = the same images again)", la, lb, None, None, 1)

imgs = [im.to(dev) for im in C.pipeline_images()]
with torch.inference_mode():
    emb = hf.embed.embedding_images({imgs[0]: ["a"], imgs[1]: ["b"], imgs[2]: ["c"]})
m = [emb[k]["mask"] for k in ("a", "b", "c")]
t2 = torch.cat([m[0], m[1]], 0); s2 = torch.cat([m[1], m[2]], 0)
with torch.inference_mode():
    hf.stages.shape_adaptor(t2, s2)
    hf.stages.shape_adaptor(torch.cat([t2, t2], 0), torch.cat([s2, s2], 0))
_, sa_ = record(lambda: hf.stages.shape_adaptor(t2, s2))
_, sb_ = record(lambda: hf.stages.shape_adaptor(torch.cat([t2, t2], 0), torch.cat([s2, s2], 0)))
compare("shape adaptor 2 vs 4 pairs", sa_, sb_, None, None, 1)
