"""A/B of the two 3x3 stride-1 f16x3 kernels on the generator's same-resolution layers (batch 8, pre-split input, split output):
conv_mfma_h<2,2,1,8,pre> (convh.hip: modulated tail, fused ToRGB) against conv_enc_h<64x512,pre> (convh_enc.hip: encoder tail).
The encoder kernel has no per-sample demodulation / noise / ToRGB slabs yet - this compares the contraction + a split-output
epilogue only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hairfastgan_amd import _marshal as M  # noqa: E402
from hairfastgan_amd._runtime import lib, stream  # noqa: E402


def timeit(fn, iters=int(os.environ.get("PROBE_ITERS", "20"))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    L, st = lib(), stream()
    B = int(os.environ.get("PROBE_BATCH", "8"))
    print("lib:", os.environ.get("HAIRFAST_HIP_LIB", "default"), "batch", B)
    for c, h in [(512, 64), (256, 128), (128, 256), (64, 512)]:
        torch.manual_seed(0)
        x = torch.randn(B, c, h, h, device=dev)
        wgt = torch.randn(1, c, c, 3, 3, device=dev)
        s, d, s2 = (torch.rand(B, c, device=dev) + 0.5 for _ in range(3))
        nz, nw, bias = torch.randn(B, 1, h, h, device=dev), torch.tensor([0.3], device=dev), torch.randn(c, device=dev)
        wt, _ = M.prepare_weights(L, st, wgt)
        hi, lo = M.split_weights_f16(L, st, wt)
        xs = M.SplitActivation(*M.split_activation_reference(x, s), None)
        gf = 2.0 * c * c * 9 * h * h * B / 1e9
        rows = []
        t = timeit(lambda: M.modconv3x3_f16_pre(L, st, xs, hi, lo, 3, d, nz, nw, bias, rgb=None, want_out=False, split_for=s2))
        rows.append(("gen split-out", t, L.hf_debug_last_path()))
        t = timeit(lambda: M.modconv3x3_f16_pre(L, st, xs, hi, lo, 3, d, nz, nw, bias, rgb=None, want_out=True, split_for=None))
        rows.append(("gen f32-out", t, L.hf_debug_last_path()))
        w2 = torch.randn(c, c, 3, 3, device=dev) / (c * 9) ** 0.5
        wte = M.conv_prepare(L, st, w2)
        ehi, elo = M.conv_split_weights_f16(L, st, wte)
        osc = torch.rand(c, device=dev) + 0.5
        xe = M.SplitActivation(xs.hi, xs.lo, None)
        if M.conv2d_f16_split_supported(L, B, c, c, h, h, 1, 3, pre=True):
            t = timeit(lambda: M.conv2d_f16_split(L, st, xe, ehi, elo, 3, c, 1, out_scale=osc, bias=bias, act=M.ACT_LRELU, alpha=0.2,
                                                   next_scale=osc, want_f32=False))
            rows.append(("enc split-out", t, L.hf_debug_last_path()))
        t = timeit(lambda: M.conv2d_f16(L, st, xe, ehi, elo, 3, c, 1, out_scale=osc, bias=bias, act=M.ACT_LRELU, alpha=0.2))
        rows.append(("enc f32-out", t, L.hf_debug_last_path()))
        for name, t, path in rows:
            print(f"{c:4d}->{c:4d} @{h:4d}  {name:14s} path {path:4d} {t:8.1f} us {gf / t * 1e3:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
