"""hf_conv1x1_f16_f32 on the GEMM shapes of a batched swap pass (CLIP tower, SEAN 1x1 convs, PostProcess shortcuts, the e4e
heads' patch GEMM), one by one on resident inputs.  HAIRFAST_HIP_LIB selects an A/B build."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hairfastgan_amd import _marshal as M  # noqa: E402
from hairfastgan_amd._runtime import lib, stream  # noqa: E402

# (label, B, cin, cout, H, W, groups, pre-split input)
SHAPES = [("CLIP qkv 768->2304", 1, 768, 2304, 64, 50, 1, True), ("CLIP fc 768->3072", 1, 768, 3072, 64, 50, 1, True),
          ("CLIP proj 3072->768", 1, 3072, 768, 64, 50, 1, True), ("SEAN 256->256 @128^2", 64, 256, 256, 128, 128, 1, False),
          ("SEAN 512->512 @64 x 19", 1, 512, 512, 64, 19, 1, True), ("pp shortcut 1024->768 @64^2", 32, 1024, 768, 64, 64, 1, True),
          ("e4e lat 256->512 @32^2", 96, 256, 512, 32, 32, 1, False), ("heads patch 4608->512 x11", 96, 4608, 512, 1, 64, 11, False)]


def main():
    dev = torch.device("cuda:0")
    L, st = lib(), stream()
    L.hf_debug_set_tuning(int(os.environ.get("PROBE_TUNE", "0")))  # 2: never the 128-channel block form (A/B)
    print("lib:", os.environ.get("HAIRFAST_HIP_LIB", "default"), "tuning", os.environ.get("PROBE_TUNE", "0"))
    for label, B, cin, cout, H, W, G, pre in SHAPES:
        torch.manual_seed(0)
        x = torch.randn(*((G, B) if G > 1 else (B,)), cin, H, W, device=dev)
        w = torch.randn(G, cout, cin, 1, 1, device=dev) / cin ** 0.5
        wt = torch.stack([M.conv_prepare(L, st, w[g]) for g in range(G)]).contiguous()
        hi, lo = M.conv_split_weights_f16(L, st, wt if G > 1 else wt[0])
        bias = torch.randn(G, cout, device=dev) if G > 1 else torch.randn(cout, device=dev)
        xin = M.split_activation_f16(L, st, x) if pre else x
        fn = lambda: M.conv1x1_f16(L, st, xin, hi, lo, 3, cout, 1, bias=bias, groups=G, x_shared=False)  # noqa: E731
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
        gf = 2.0 * cin * cout * H * W * B * G / 1e9
        print(f"{label:30s} path {L.hf_debug_last_path()} {gf:8.1f} GFLOP {us:9.1f} us {gf / us * 1e3:7.1f} TFLOP/s = {gf / us * 1e3 / 8.389:4.1f} % of the f16x3 roof", flush=True)


if __name__ == "__main__":
    main()
