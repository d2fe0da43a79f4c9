#!/usr/bin/env python3
"""Per-layer micro-benchmark of the modulated-conv kernels on the GPU box: every 3x3 layer
shape of Generator(1024) at a given batch, under each tile configuration
(hf_debug_set_dispatch), timed with HIP events.  Prints TFLOP/s (algorithmic FLOPs) and
GB/s for the streaming kernels.  Usage: python tools/bench_layers.py [--batch 8] [--iters 10]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hairfastgan_amd import _marshal as M  # noqa: E402
from hairfastgan_amd._runtime import lib, stream  # noqa: E402


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--same", default="0,300,354")
    ap.add_argument("--up", default="0,300,100")
    ap.add_argument("--only", default="", help="restrict to e.g. 'same:64,up:128' (kind:input resolution)")
    ap.add_argument("--no-streaming", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B = args.batch
    L = lib()
    ch = {4: 512, 8: 512, 16: 512, 32: 512, 64: 512, 128: 256, 256: 128, 512: 64, 1024: 32}
    same_layers = [(ch[r], ch[r], r) for r in (4, 8, 16, 32, 64, 128, 256, 512, 1024)]
    up_layers = [(ch[r], ch[2 * r], r) for r in (4, 8, 16, 32, 64, 128, 256, 512)]
    print(f"batch {B}; TFLOP/s per config (fp32 MFMA peak 157.3)")
    for up, layers, cfgs in ((False, same_layers, args.same), (True, up_layers, args.up)):
        cfgs = [int(c) for c in cfgs.split(",") if c]
        print(("UP  " if up else "SAME") + " cin  cout  res | " + " ".join(f"{c:>6d}" for c in cfgs))
        for cin, cout, r in layers:
            if args.only and f"{'up' if up else 'same'}:{r}" not in args.only.split(","):
                continue
            x = torch.randn(B, cin, r, r, device=dev)
            wgt = torch.randn(1, cout, cin, 3, 3, device=dev)
            wt, wsq = M.prepare_weights(L, stream(), wgt)
            s = torch.rand(B, cin, device=dev) + 0.5
            d = torch.rand(B, cout, device=dev) + 0.5
            oh = 2 * r if up else r
            noise = torch.randn(1, 1, oh, oh, device=dev)
            nw = torch.tensor([0.1], device=dev)
            bias = torch.randn(cout, device=dev)
            flops = 2.0 * cin * cout * 9 * r * r * B
            ws_n = L.hf_modconv_workspace_floats(B, cin, cout, r, r, 1 if up else 0)
            ws = torch.empty(max(ws_n, 1), device=dev)
            ws_p = ws.data_ptr() if ws_n else None
            row = []
            for c in cfgs:
                L.hf_debug_set_dispatch(0 if (up or c >= 100) else c, c if (up and c < 100) else 0)
                try:
                    if up and c >= 100:  # fp16 matrix cores: 100 = plain f16, 300 = split operands
                        if not M.modconv3x3_up_f16_supported(cin, cout, r, r):
                            row.append("     -")
                            continue
                        hi, lo = M.split_weights_f16(L, stream(), wt)
                        pitch = L.hf_modconv_up_pitch(r)
                        tmp = torch.empty(B, cout, 2 * r + 1, pitch, device=dev)

                        def fn(nt=c // 100):
                            code = L.hf_modconv3x3_up_f16_f32(tmp.data_ptr(), x.data_ptr(), hi.data_ptr(), lo.data_ptr(), nt,
                                                              s.data_ptr(), d.data_ptr(), B, cin, cout, r, r, pitch, stream())
                            assert code == 0, code
                    elif up:
                        pitch = L.hf_modconv_up_pitch(r)
                        tmp = torch.empty(B, cout, 2 * r + 1, pitch, device=dev)
                        fn = lambda: L.hf_modconv3x3_up_f32(tmp.data_ptr(), x.data_ptr(), wt.data_ptr(), s.data_ptr(),
                                                            d.data_ptr(), B, cin, cout, r, r, pitch, ws_p, ws_n, stream())
                    elif c in (400, 451, 452):  # fp16 matrix cores, pre-split K-blocked input (hf_modconv3x3_f16_pre_f32); 45x = forced tile
                        if not M.modconv3x3_f16_supported(cin, cout, r, r):
                            row.append("     -")
                            continue
                        hi, lo = M.split_weights_f16(L, stream(), wt)
                        xh, xl = M.split_activation_reference(x, s)
                        out = torch.empty(B, cout, r, r, device=dev)
                        L.hf_debug_set_dispatch(c - 400 if c != 400 else 0, 0)

                        def fn():
                            code = L.hf_modconv3x3_f16_pre_f32(out.data_ptr(), xh.data_ptr(), xl.data_ptr(), hi.data_ptr(),
                                                               lo.data_ptr(), 3, d.data_ptr(), noise.data_ptr(), nw.data_ptr(), 0,
                                                               bias.data_ptr(), B, cin, cout, r, r, 0.2, 1.4142135, None, None,
                                                               None, None, None, None, stream())
                            assert code == 0, code
                    elif c >= 100:  # fp16 matrix cores (csrc/convh.hip): 1TT = plain f16, 3TT = split operands; TT = tile cfg (00 auto)
                        if not M.modconv3x3_f16_supported(cin, cout, r, r):
                            row.append("     -")
                            continue
                        hi, lo = M.split_weights_f16(L, stream(), wt)
                        out = torch.empty(B, cout, r, r, device=dev)

                        L.hf_debug_set_dispatch(c % 100, 0)

                        def fn(nt=c // 100):
                            code = L.hf_modconv3x3_f16_f32(out.data_ptr(), x.data_ptr(), hi.data_ptr(), lo.data_ptr(), nt,
                                                           s.data_ptr(), d.data_ptr(), noise.data_ptr(), nw.data_ptr(), 0,
                                                           bias.data_ptr(), B, cin, cout, r, r, 0.2, 1.4142135, stream())
                            assert code == 0, code
                    else:
                        out = torch.empty(B, cout, r, r, device=dev)
                        fn = lambda: L.hf_modconv3x3_f32(out.data_ptr(), x.data_ptr(), wt.data_ptr(), s.data_ptr(),
                                                         d.data_ptr(), noise.data_ptr(), nw.data_ptr(), 0,
                                                         bias.data_ptr(), B, cin, cout, r, r, 0.2, 1.4142135,
                                                         ws_p, ws_n, stream())
                    t = timeit(fn, args.iters)
                    row.append(f"{flops / t / 1e12:6.1f}")
                except Exception as e:  # noqa: BLE001
                    row.append("   err")
                finally:
                    L.hf_debug_set_dispatch(0, 0)
            print(f"     {cin:4d} {cout:4d} {r:5d} | " + " ".join(row), flush=True)
    if args.no_streaming:
        return
    # streaming kernels: blur+noise+act and ToRGB at the top resolutions
    print("streaming kernels (GB/s, algorithmic bytes = read input once + write output once)")
    k4 = torch.tensor([1., 3., 3., 1.], device=dev)
    k4 = (k4[None] * k4[:, None]) / 64 * 4
    for c, r in ((32, 1024), (64, 512), (128, 256), (256, 128)):
        pitch = (r + 1 + 3) & ~3
        tmp = torch.randn(B, c, r + 1, pitch, device=dev)
        noise = torch.randn(1, 1, r, r, device=dev)
        nw = torch.tensor([0.1], device=dev)
        bias = torch.randn(c, device=dev)
        t = timeit(lambda: M.noise_bias_act(L, stream(), tmp, None, nw, bias) if False else
                   L.hf_blur_noise_bias_act_f32(out_b.data_ptr(), tmp.data_ptr(), k4.data_ptr(), noise.data_ptr(),
                                                nw.data_ptr(), 0, bias.data_ptr(), B, c, r + 1, r + 1, pitch, 0.2, 1.4142,
                                                stream()), args.iters) if (out_b := torch.empty(B, c, r, r, device=dev)) is not None else 0
        byts = 4.0 * B * c * ((r + 1) ** 2 + r * r)
        print(f"  blur+noise+act  c={c:4d} out={r:5d}: {t * 1e6:8.1f} us  {byts / t / 1e9:7.0f} GB/s")
        xh = torch.empty(B, c // 8, r, r, 8, dtype=torch.float16, device=dev)
        xl = torch.empty_like(xh)
        sn = torch.rand(B, c, device=dev)
        t = timeit(lambda: L.hf_blur_noise_bias_act_split_f16(xh.data_ptr(), xl.data_ptr(), tmp.data_ptr(), k4.data_ptr(),
                                                              noise.data_ptr(), nw.data_ptr(), 0, bias.data_ptr(), sn.data_ptr(),
                                                              B, c, r + 1, r + 1, pitch, 0.2, 1.4142, stream()), args.iters)
        print(f"  blur -> split16 c={c:4d} out={r:5d}: {t * 1e6:8.1f} us  {byts / t / 1e9:7.0f} GB/s")
        x = torch.randn(B, c, r, r, device=dev)
        wrgb = torch.randn(1, c, 3, device=dev)
        s = torch.rand(B, c, device=dev)
        b3 = torch.randn(3, device=dev)
        skip = torch.randn(B, 3, r // 2, r // 2, device=dev)
        out = torch.empty(B, 3, r, r, device=dev)
        t = timeit(lambda: L.hf_torgb_f32(out.data_ptr(), x.data_ptr(), wrgb.data_ptr(), s.data_ptr(), b3.data_ptr(),
                                          skip.data_ptr(), k4.data_ptr(), B, c, r, r, stream()), args.iters)
        byts = 4.0 * B * (c * r * r + 3 * r * r + 3 * r * r / 4)
        print(f"  torgb           c={c:4d} res={r:5d}: {t * 1e6:8.1f} us  {byts / t / 1e9:7.0f} GB/s")


if __name__ == "__main__":
    main()
