"""Where the device-to-device copies of ONE generator forward (batch 8) come from: torch.profiler with Python stacks,
copy / elementwise launches grouped by the innermost hairfastgan_amd frame."""
import collections, os, sys
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
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], with_stack=True) as prof:
        g([lat], input_is_latent=True)
        torch.cuda.synchronize()
tally = collections.Counter()
for e in prof.events():
    if e.device_type.name == "CPU" and e.name in ("aten::copy_", "aten::contiguous", "aten::clone", "aten::fill_", "aten::zero_", "aten::cat",
                                                   "aten::mul", "aten::add", "aten::to", "aten::_to_copy", "aten::empty_strided"):
        fr = [s for s in (e.stack or []) if "hairfastgan_amd" in s]
        tally[(e.name, fr[0] if fr else "?")] += 1
for (name, where), n in tally.most_common(40):
    print(f"{n:4d} {name:22s} {where}")
