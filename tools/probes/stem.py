"""Timing of the fused ResNet stem (csrc/stem.hip) against the general conv + max-pool pair it replaces."""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from hairfastgan_amd import _marshal as M  # noqa: E402
from hairfastgan_amd._runtime import lib, stream  # noqa: E402


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


dev = torch.device("cuda:0")
L, st = lib(), stream()
w = torch.randn(64, 3, 7, 7, device=dev) * 0.05
sc, sh = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.3
wt = M.conv_prepare(L, st, w)
w3 = M.stem_prepare(w)
for (B, S) in [(16, 1024), (24, 512), (2, 1024), (3, 512)]:
    x = torch.randn(B, 3, S, S, device=dev)
    old = timeit(lambda: M.maxpool3x3s2(L, st, M.conv2d(L, st, x, wt, 7, 2, out_scale=sc, bias=sh, act=M.ACT_LRELU, alpha=0.0)))
    new = timeit(lambda: M.stem7x7s2(L, st, x, w3, out_scale=sc, bias=sh, alpha=0.0, pool=True))
    plain = timeit(lambda: M.stem7x7s2(L, st, x, w3, out_scale=sc, bias=sh, alpha=0.0, pool=False))
    gf = 2.0 * 147 * 64 * (S // 2) ** 2 * B / 1e9
    print(f"B={B} {S}^2: conv+pool {old:8.1f} us | fused stem {new:8.1f} us ({gf / new * 1e3:6.1f} TF/s useful) | stem without pool {plain:8.1f} us", flush=True)
