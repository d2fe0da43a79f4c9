"""A swap in the default mode, then the same swap with the process switched to f32 and to f16 AFTER the plans were built
(what HAIRFAST_CONV_PRECISION=auto's re-run does): every route must still work on plans prepared for another mode."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from hairfastgan_amd import _runtime

dev = torch.device("cuda:0")
_g, sd = bench.build_generator(dev)
hf = bench.build_hairfast(sd, dev)
load = bench.make_triple_loader(2)
trip = [t.to(dev) for t in load(0)]
with torch.inference_mode():
    ref = hf.swap(*trip).float()
    for mode in ("f32", "f16", "f16x3"):
        prev = _runtime.set_conv_precision(mode)
        try:
            out = hf.swap(*trip).float()
            two = hf.swap_batch([tuple(trip), tuple(t.to(dev) for t in load(1))])
        finally:
            _runtime.set_conv_precision(prev)
        d = (out - ref).abs()
        print(f"mode {mode}: max |diff| vs default {float(d.max()):.1f} / 255, mean {float(d.mean()):.4f}; batch of 2 ok {len(two)}", flush=True)
