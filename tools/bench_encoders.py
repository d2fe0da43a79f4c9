#!/usr/bin/env python3
"""Time the two encoder forwards on the GPU box (synthetic weights): e4e on [B,3,256,256],
FS encoder on [B,3,1024,1024] (downscale + trunk + heads), B = 1, 2, 3 (the batches HairFast
uses, Embedding.py:51,71,74).  Reports ms and TFLOP/s (145.0 / 69.6 GFLOP per image)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hairfastgan_amd.encoders import Encoder4Editing, FSEncoder  # noqa: E402
from oracle import cases as C  # noqa: E402
from oracle import ref_encoders as E  # noqa: E402


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda:0")
    e4e = Encoder4Editing(50, "ir_se", argparse.Namespace(stylegan_size=1024)).eval()
    e4e.load_state_dict(C.params_from_shapes("e4e", E.e4e_param_shapes()))
    e4e = e4e.to(dev)
    fs = FSEncoder()
    fs.enc.load_state_dict(C.params_from_shapes("fs", E.fs_param_shapes()))
    fs = fs.to(dev)
    from hairfastgan_amd import _runtime
    from hairfastgan_amd.graphs import GraphRunner

    modes = sys.argv[1:] or ["f16x3"]
    for mode in modes:
        _runtime.set_conv_precision(mode)
        print(f"--- conv precision {mode}", flush=True)
        with torch.inference_mode():
            for B in (1, 2, 3, 8):
                x = torch.randn(B, 3, 256, 256, device=dev)
                img = torch.randn(B, 3, 1024, 1024, device=dev)
                t1 = timeit(lambda: e4e(x))
                t2 = timeit(lambda: fs.test(img=img, return_latent=True))
                line = (f"B={B}: e4e {t1:7.2f} ms ({145.0 * B / t1:6.1f} TFLOP/s)   fs-encoder {t2:7.2f} ms "
                        f"({69.6 * B / t2:6.1f} TFLOP/s)")
                if B <= 3:  # hipGraph replay: the launch-bound regime
                    g1 = GraphRunner(lambda a: e4e(a), x)
                    g2 = GraphRunner(lambda a: fs.test(img=a, return_latent=True)[2:], img)
                    line += f"   hipGraph: e4e {timeit(lambda: g1(x)):6.2f} ms  fs {timeit(lambda: g2(img)):6.2f} ms"
                print(line, flush=True)

if __name__ == "__main__":
    main()
