"""GPU launches of one shape-adaptor call (B pairs) by kernel name, from torch.profiler."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch.profiler import profile, ProfilerActivity
from hairfastgan_amd.shape_adaptor import MaskGenerator, adapt_shape
from oracle import cases as C

dev = torch.device("cuda:0")
gen = MaskGenerator().eval(); gen.load_state_dict(C.shape_adaptor_params()); gen.to(dev)
m1, m2 = (m.to(dev) for m in C.shape_masks())
B = int(os.environ.get("PROBE_BATCH", "2"))
m1, m2 = m1.repeat(B // 2 + 1, 1, 1, 1)[:B], m2.repeat(B // 2 + 1, 1, 1, 1)[:B]
for _ in range(2):
    adapt_shape(gen, m1, m2)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    adapt_shape(gen, m1, m2)
    torch.cuda.synchronize()
rows = [(e.key, e.count, e.device_time_total) for e in prof.key_averages() if e.device_time_total > 0 and e.device_type.name != "CPU"]
rows.sort(key=lambda r: -r[2])
print(f"B={B}: total device time {sum(r[2] for r in rows):.0f} us, {sum(r[1] for r in rows)} launches")
for k, n, t in rows[:22]:
    print(f"{t:9.1f} us {n:4d}x {t / n:8.1f}  {k[:100]}")
