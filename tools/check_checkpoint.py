#!/usr/bin/env python3
"""Validate the default conv precision (f16x3) on a REAL StyleGAN2 checkpoint - one command for the first person who has
weights (none exist in the build container or on the test box: every number in this repository comes from synthetic
closed-form parameters).

    python tools/check_checkpoint.py pretrained_models/StyleGAN/ffhq.pt [--samples 8] [--truncation 0.7] [--seed 0]

Loads ckpt['g_ema'] (+ ckpt['latent_avg']) like models/Net.py:29-46, draws z, maps it to W+ with the checkpoint's own
mapping network, and runs the 1024^2 generator
  * in f32 mode with hooks: per layer the largest |activation| and the largest modulation |s| - the quantities the fp16
    (hi, lo) split's range depends on (the split carries |s*x| < 131008 after the style normalisation to max|s| in [1,2));
  * in f16x3 mode: the library's clamp counter (hf_f16_overflow_count; must be 0) and the image difference to f32
    (max-abs, MSE, PSNR on [-1,1] images) - expected: fp32-class, max-abs ~1e-5 of the image range;
  * in f16 mode (BASELINE.json configs[4]) for comparison.
Exit status 1 if anything clamped or the f16x3 image differs from the f32 one by more than 1e-3 (then run with
HAIRFAST_CONV_PRECISION=auto or f32 and report the layer statistics).

    python tools/check_checkpoint.py --swap [--pretrained-root DIR] [--images face.npy shape.npy color.npy]

The WHOLE swap on the reference's checkpoint files (round 4: `HairFast(args)` reads them like the reference's constructor -
hairfastgan_amd/checkpoints.py; run it from the HairFastGAN checkout or pass --pretrained-root): one swap of three images
(HWC uint8 .npy arrays; seeded random images without --images - enough to exercise every layer's range, not a meaningful
picture) in f32, f16x3 and f16 mode with the same seed: clamp counter, final-image difference to the f32 run, and whether
every segmentation mask index of the two runs agrees (the masks gate the argmax-dependent stages).  Same exit rule."""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def check_swap(args):
    import numpy as np

    from hairfastgan_amd import _marshal, _runtime
    from hairfastgan_amd import hair_swap as HS

    assert torch.cuda.is_available(), "needs the MI355X (no CPU fallback)"
    hargs = HS.get_parser().parse_args([])
    hargs.device = torch.device("cuda:0")
    hf = HS.HairFast(hargs, pretrained_root=args.pretrained_root)  # FileNotFoundError names a missing checkpoint
    if args.images:
        imgs = [torch.from_numpy(np.load(p_)).permute(2, 0, 1).contiguous() for p_ in args.images]
    else:
        g = torch.Generator().manual_seed(args.seed)
        imgs = [torch.randint(0, 256, (3, 1024, 1024), dtype=torch.uint8, generator=g) for _ in range(3)]
    out, masks = {}, {}
    seg = HS.get_segmentation
    for mode in ("f32", "f16x3", "f16"):
        rec = []
        HS.get_segmentation = lambda net, x, **kw: (rec.append(seg(net, x, **kw)) or rec[-1])
        hf.conv_precision = mode
        try:
            _marshal.f16_overflow_count(_runtime.lib(), reset=True)
            img = hf.swap(*imgs, seed=args.seed).float()
            out[mode] = (img, _marshal.f16_overflow_count(_runtime.lib()))
            masks[mode] = [m.clone() for m in rec]
        finally:
            HS.get_segmentation = seg
    ref = out["f32"][0]
    bad = False
    for mode in ("f16x3", "f16"):
        img, clamped = out[mode]
        err, mse = (img - ref).abs(), float((img - ref).pow(2).mean())
        flips = sum(int((a != b).sum()) for a, b in zip(masks[mode], masks["f32"]))
        print(f"{mode:6s}: clamped elements {clamped}, final image max-abs vs f32 {float(err.max()):.3e} (range [0,1]), MSE {mse:.3e}, "
              f"mask indices differing from the f32 run: {flips} of {sum(m.numel() for m in masks['f32'])}")
        if mode == "f16x3" and (clamped > 0 or float(err.max()) > 1e-3):
            bad = True
    print("f16x3 is " + ("NOT safe as the default for this swap: use HAIRFAST_CONV_PRECISION=auto or f32" if bad
                         else "safe on this swap (fp32-class final image, nothing clamped)"))
    return 1 if bad else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("ckpt", nargs="?")
    ap.add_argument("--swap", action="store_true", help="check one whole HairFast.swap on the reference's checkpoint files")
    ap.add_argument("--pretrained-root", default=None)
    ap.add_argument("--images", nargs=3, default=None, help="face, shape, color as HWC uint8 .npy arrays")
    ap.add_argument("--samples", type=int, default=8)
    ap.add_argument("--truncation", type=float, default=1.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--n-mlp", type=int, default=8)
    ap.add_argument("--channel-multiplier", type=int, default=2)
    args = ap.parse_args()
    if args.swap:
        return check_swap(args)
    if not args.ckpt:
        ap.error("a StyleGAN2 checkpoint path (or --swap)")
    from hairfastgan_amd import _marshal, _runtime
    from hairfastgan_amd.stylegan2.model import Generator, StyledConv

    assert torch.cuda.is_available(), "needs the MI355X (no CPU fallback)"
    dev = torch.device("cuda:0")
    ckpt = torch.load(args.ckpt, map_location="cpu")
    g = Generator(args.size, 512, args.n_mlp, channel_multiplier=args.channel_multiplier).eval()
    g.load_state_dict(ckpt["g_ema"])
    g = g.to(dev)
    torch.manual_seed(args.seed)
    z = torch.randn(args.samples, 512, device=dev)
    with torch.inference_mode():
        w = g.style(z)
        if args.truncation < 1:
            mean_w = ckpt["latent_avg"].to(dev) if "latent_avg" in ckpt else g.mean_latent(4096)
            w = mean_w + args.truncation * (w - mean_w)
        latent = w.unsqueeze(1).repeat(1, g.n_latent, 1)
        noise = [n.to(dev) for n in g.make_noise()]
        stats, hooks = {}, []
        for name, m in g.named_modules():
            if isinstance(m, StyledConv):
                hooks.append(m.register_forward_hook(
                    lambda mod, inp, out, name=name: stats.__setitem__(name, (float(inp[0].abs().max()) if torch.is_tensor(inp[0]) else float("nan"),
                                                                              float(out.abs().max()) if torch.is_tensor(out) else float("nan")))))
        images = {}
        for mode in ("f32", "f16x3", "f16"):
            prev = _runtime.set_conv_precision(mode)
            try:
                _marshal.f16_overflow_count(_runtime.lib(), reset=True)
                img, _ = g([latent], input_is_latent=True, noise=noise)
                images[mode] = (img.float(), _marshal.f16_overflow_count(_runtime.lib()))
            finally:
                _runtime.set_conv_precision(prev)
            if mode == "f32":
                for h in hooks:
                    h.remove()
    print(f"checkpoint {args.ckpt}: {args.samples} samples, truncation {args.truncation}")
    print("per StyledConv (f32 run): max |input|, max |output|")
    for name, (a, b) in stats.items():
        print(f"  {name:12s} {a:12.4g} {b:12.4g}" + ("   <-- beyond the fp16-pair range (65504) before modulation" if a > 65504 else ""))
    ref = images["f32"][0]
    bad = False
    for mode in ("f16x3", "f16"):
        img, clamped = images[mode]
        err = (img - ref).abs()
        mse = float((img - ref).pow(2).mean())
        psnr = 10 * math.log10(4.0 / mse) if mse > 0 else float("inf")
        print(f"{mode:6s}: clamped elements {clamped}, max-abs vs f32 {float(err.max()):.3e}, MSE {mse:.3e}, PSNR {psnr:.1f} dB")
        if mode == "f16x3" and (clamped > 0 or float(err.max()) > 1e-3):
            bad = True
    print("f16x3 is " + ("NOT safe as the default on this checkpoint: use HAIRFAST_CONV_PRECISION=auto (re-runs clamped forwards in f32) or f32"
                         if bad else "safe on these samples (fp32-class images, nothing clamped)"))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
