"""Duration of each of the first forwards of a fresh process (generator 0->8, batch 8): how many steps the chip / the allocator
need to reach the steady state bench.py's long runs report."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hairfastgan_amd.stylegan2.model import Generator

dev = torch.device("cuda:0")
torch.manual_seed(0)
g = Generator(1024, 512, 8, channel_multiplier=2).to(dev).eval()
lat = torch.randn(8, 18, 512, device=dev)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
with torch.inference_mode():
    ev[0].record()
    for i in range(40):
        g([lat], input_is_latent=True)
        ev[i + 1].record()
torch.cuda.synchronize()
print("ms per forward:", " ".join(f"{ev[i].elapsed_time(ev[i + 1]):.2f}" for i in range(40)))
print("reserved MB", torch.cuda.memory_reserved() >> 20)
