#!/usr/bin/env python3
"""Per-layer timing of the encoders' 3x3 convs on the GPU box: fp32 MFMA (hf_conv2d_f32) vs the fp16
matrix-core kernel (hf_conv2d_f16_f32: f16x3 with in-kernel conversion, f16x3 on a pre-split input incl.
the split pass, f16).  Shapes: the e4e / FS-encoder / PostProcess layers at the batch sizes HairFast uses."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hairfastgan_amd import _marshal as M  # noqa: E402
from hairfastgan_amd._runtime import lib, stream  # noqa: E402

# (label, B, cin, cout, H, W, stride, groups)
LAYERS = [
    ("e4e/fs 64@256^2", 3, 64, 64, 256, 256, 1, 1), ("e4e/fs 64@256^2 s2", 3, 64, 64, 256, 256, 2, 1),
    ("e4e 64@128^2", 3, 64, 64, 128, 128, 1, 1), ("e4e 64->128@128^2", 3, 64, 128, 128, 128, 1, 1),
    ("e4e 128@128^2 s2", 3, 128, 128, 128, 128, 2, 1), ("e4e 128@64^2", 3, 128, 128, 64, 64, 1, 1),
    ("e4e 128->256@64^2", 3, 128, 256, 64, 64, 1, 1), ("e4e 256@64^2 s2", 3, 256, 256, 64, 64, 2, 1),
    ("e4e 256@32^2", 3, 256, 256, 32, 32, 1, 1), ("e4e 256@32^2 B2", 2, 256, 256, 32, 32, 1, 1),
    ("e4e 256->512@32^2", 3, 256, 512, 32, 32, 1, 1), ("e4e 512@32^2 s2", 3, 512, 512, 32, 32, 2, 1),
    ("e4e 512@16^2", 3, 512, 512, 16, 16, 1, 1),
    ("heads fine L1 x11", 3, 512, 512, 64, 64, 2, 11), ("heads fine L2 x11", 3, 512, 512, 32, 32, 2, 11),
    ("heads mid L1 x4", 3, 512, 512, 32, 32, 2, 4),
    ("pp 1024@64^2", 1, 1024, 1024, 64, 64, 1, 1), ("pp 768@64^2", 1, 768, 768, 64, 64, 1, 1), ("pp 128->512@64^2 B2", 2, 128, 512, 64, 64, 1, 1),
]


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    dev = torch.device("cuda:0")
    L, st = lib(), stream()
    print("rooflines: fp32 MFMA 157.3 TFLOP/s, f16x3 (3 fp16 MFMAs per product) 838.9, f16 2516.6; `pre` includes the split pass")
    print(f"{'layer':24s} {'GFLOP':>7s} | {'f32 us':>8s} {'TF/s':>6s} {'%':>3s} | {'f16x3 us':>8s} {'TF/s':>6s} {'%':>3s} | {'pre us':>8s} {'TF/s':>6s} {'%':>3s} | {'f16 us':>8s} {'TF/s':>6s} {'%':>3s} | split-K")
    L.hf_debug_set_tuning(int(os.environ.get("ENC_TUNE", "0")))  # 4 = without the 512-pixel tile form
    mult = int(os.environ.get("ENC_BATCH_MULT", "1"))  # swap_batch: the same layers with `mult` triples per pass
    for label, B, cin, cout, H, W, stride, G in LAYERS:
        B *= mult
        torch.manual_seed(0)
        x = torch.randn(B, cin, H, W, device=dev)
        w = torch.randn(G, cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
        wt = torch.stack([M.conv_prepare(L, st, w[g]) for g in range(G)]).contiguous()
        if G == 1:
            wt = wt[0]
        hi, lo = M.conv_split_weights_f16(L, st, wt)
        bias = torch.randn(G, cout, device=dev) if G > 1 else torch.randn(cout, device=dev)
        kw = dict(bias=bias, act=M.ACT_LRELU, alpha=0.01, groups=G)
        oh, ow = (H - 1) // stride + 1, (W - 1) // stride + 1
        gf = 2.0 * cin * cout * 9 * oh * ow * B * G / 1e9
        t32 = timeit(lambda: M.conv2d(L, st, x, wt, 3, stride, **kw))
        t3 = timeit(lambda: M.conv2d_f16(L, st, x, hi, lo, 3, cout, stride, **kw))
        tp = timeit(lambda: M.conv2d_f16(L, st, M.split_activation_f16(L, st, x), hi, lo, 3, cout, stride, **kw))
        t1 = timeit(lambda: M.conv2d_f16(L, st, x, hi, lo, 1, cout, stride, **kw))
        sk = L.hf_conv2d_f16_workspace_floats(B, cin, cout, H, W, stride, G) // (G * B * cout * oh * ow)
        tf = lambda t: gf / t * 1e3  # noqa: E731
        print(f"{label:24s} {gf:7.2f} | {t32:8.1f} {tf(t32):6.1f} {tf(t32) / 1.573:3.0f} | {t3:8.1f} {tf(t3):6.1f} {tf(t3) / 8.389:3.0f} | "
              f"{tp:8.1f} {tf(tp):6.1f} {tf(tp) / 8.389:3.0f} | {t1:8.1f} {tf(t1):6.1f} {tf(t1) / 25.166:3.0f} | {sk}", flush=True)


if __name__ == "__main__":
    main()
