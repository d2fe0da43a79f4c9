"""Every GPU launch of ONE generator forward (batch 8, range 0->8), by name, from torch.profiler - to spot
stray torch-native launches (copies, fills, RNG) between the library's kernels."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hairfastgan_amd.stylegan2.model import Generator
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
torch.manual_seed(0)
g = Generator(1024, 512, 8, channel_multiplier=2).to(dev).eval()
B = int(os.environ.get("PROBE_BATCH", "8"))
lat = torch.randn(B, 18, 512, device=dev)
with torch.inference_mode():
    for _ in range(2):
        g([lat], input_is_latent=True)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        g([lat], input_is_latent=True)
        torch.cuda.synchronize()
rows = [(e.key, e.count, e.device_time_total) for e in prof.key_averages() if e.device_time_total > 0 and e.device_type.name != "CPU"]
if not rows:
    rows = [(e.key, e.count, e.device_time_total) for e in prof.key_averages() if e.device_time_total > 0]
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
print(f"total device time {tot:.0f} us, {sum(r[1] for r in rows)} launches")
for k, n, t in rows[:45]:
    print(f"{t:9.1f} us {n:4d}x  {k[:110]}")
