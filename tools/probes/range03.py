"""Run Generator range 0->3 a few times at a given batch (for rocprofv3): python range03.py [batch]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
g, _ = bench.build_generator(dev)
lat = torch.randn(B, 18, 512, device=dev)
with torch.inference_mode():
    for _ in range(2):
        g([lat], input_is_latent=True, end_layer=3)
    torch.cuda.synchronize()
    torch.cuda.nvtx.range_push("timed") if False else None
    for _ in range(10):
        g([lat], input_is_latent=True, end_layer=3)
    torch.cuda.synchronize()
