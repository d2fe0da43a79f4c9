"""The pre-split encoder convs of a batched swap pass (32 triples: e4e / FS encoder at batch 96, PostProcess at 64), timed one by
one on resident, already split inputs: the kernel alone (conv_enc_h<...,pre>).  HAIRFAST_HIP_LIB selects an A/B build."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hairfastgan_amd import _marshal as M  # noqa: E402
from hairfastgan_amd._runtime import lib, stream  # noqa: E402

LAYERS = [("e4e 64@128^2", 96, 64, 64, 128, 128, 1), ("e4e 128@64^2", 96, 128, 128, 64, 64, 1), ("e4e 256@32^2", 96, 256, 256, 32, 32, 1),
          ("e4e 512@16^2", 96, 512, 512, 16, 16, 1), ("fs 64@256^2 s2", 48, 64, 64, 256, 256, 2), ("e4e 128@128^2 s2", 96, 128, 128, 128, 128, 2),
          ("e4e 256@64^2 s2", 96, 256, 256, 64, 64, 2), ("e4e 512@32^2 s2", 96, 512, 512, 32, 32, 2), ("pp 1024@64^2", 32, 1024, 1024, 64, 64, 1),
          ("pp 512@64^2", 64, 512, 512, 64, 64, 1), ("e4e 256@32^2 B3", 3, 256, 256, 32, 32, 1), ("e4e 128@64^2 B3", 3, 128, 128, 64, 64, 1),
          ("heads L1 x11 s2", 96, 512, 512, 64, 64, 2, 11), ("heads L1 x4 s2", 96, 512, 512, 32, 32, 2, 4)]


def main():
    dev = torch.device("cuda:0")
    L, st = lib(), stream()
    print("lib:", os.environ.get("HAIRFAST_HIP_LIB", "default"))
    for label, B, cin, cout, H, W, stride, *rest in LAYERS:
        G = rest[0] if rest else 1
        torch.manual_seed(0)
        x = torch.randn(B, cin, H, W, device=dev)
        w = torch.randn(G, cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
        wt = torch.stack([M.conv_prepare(L, st, w[g]) for g in range(G)]).contiguous()
        hi, lo = M.conv_split_weights_f16(L, st, wt if G > 1 else wt[0])
        bias = torch.randn(G, cout, device=dev) if G > 1 else torch.randn(cout, device=dev)
        xs = M.split_activation_f16(L, st, x)
        fn = lambda: M.conv2d_f16(L, st, xs, hi, lo, 3, cout, stride, bias=bias, act=M.ACT_LRELU, alpha=0.01, groups=G)  # noqa: E731
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        oh, ow = (H - 1) // stride + 1, (W - 1) // stride + 1
        gf = 2.0 * cin * cout * 9 * oh * ow * B * G / 1e9
        print(f"{label:20s} B={B:3d} path {L.hf_debug_last_path()} {gf:8.1f} GFLOP {us:9.1f} us {gf / us * 1e3:7.1f} TFLOP/s = {gf / us * 1e3 / 8.389:4.1f} % of the f16x3 roof", flush=True)


if __name__ == "__main__":
    main()
