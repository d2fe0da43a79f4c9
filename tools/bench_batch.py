#!/usr/bin/env python3
"""Generator forward time vs batch size (the swap schedule runs the generator at batch 1-3) and per
layer range; default conv precision.  usage: python tools/bench_batch.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    g, _ = bench.build_generator(dev)
    with torch.inference_mode():
        for B in (1, 2, 3, 4, 8):
            lat = torch.randn(B, 18, 512, device=dev)
            for rng, kw in (("0->8", {}), ("0->3", {"end_layer": 3}),
                            ("4->8", {"start_layer": 4, "layer_in": torch.randn(B, 512, 32, 32, device=dev)}),
                            ("5->8", {"start_layer": 5, "layer_in": torch.randn(B, 512, 64, 64, device=dev)})):
                for _ in range(3):
                    g([lat], input_is_latent=True, **kw)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = 20
                for _ in range(n):
                    g([lat], input_is_latent=True, **kw)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n
                print(f"B={B} range {rng}: {dt * 1e3:7.3f} ms  ({dt / B * 1e3:6.3f} ms/img)", flush=True)


if __name__ == "__main__":
    main()
