"""GPU launches of ONE e4e / FS-encoder forward (batch 3) by kernel name, from torch.profiler."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
g, sd = bench.build_generator(dev)
hf = bench.build_hairfast(sd, dev)
B = int(os.environ.get("PROBE_BATCH", "3"))
x = torch.randn(B, 3, 256, 256, device=dev)
x1024 = torch.randn(B, 3, 1024, 1024, device=dev)
from hairfastgan_amd.encoders import get_latents
for name, fn in (("e4e", lambda: get_latents(hf.embed.e4e, x)), ("fs", lambda: hf.embed.encoder.test(img=x1024, return_latent=True))):
    with torch.inference_mode():
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
            fn()
            torch.cuda.synchronize()
    rows = [(e.key, e.count, e.device_time_total) for e in prof.key_averages() if e.device_time_total > 0 and e.device_type.name != "CPU"]
    rows.sort(key=lambda r: -r[2])
    print(f"== {name} B={B}: total device time {sum(r[2] for r in rows):.0f} us, {sum(r[1] for r in rows)} launches")
    for k, n, t in rows[:18]:
        print(f"{t:9.1f} us {n:4d}x {t / n:7.1f}  {k[:100]}")
