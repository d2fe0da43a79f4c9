"""Timing of the 1024^2 32->32 layer (csrc/convrow.hip) with parts of the kernel switched off (development probe)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from hairfastgan_amd import _marshal as M
from hairfastgan_amd._runtime import lib, stream
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
dev = torch.device("cuda:0"); L, st = lib(), stream()
B, H = 8, 1024
x = torch.randn(B, 32, H, H, device=dev)
wgt = torch.randn(1, 32, 32, 3, 3, device=dev)
s, dm = torch.rand(B, 32, device=dev) + 0.5, torch.rand(B, 32, device=dev) + 0.5
nz, nw, bias = torch.randn(B, 1, H, H, device=dev), torch.tensor([0.3], device=dev), torch.randn(32, device=dev)
rgb_w, rgb_s = torch.randn(32, 3, device=dev) * 0.2, torch.rand(B, 32, device=dev) + 0.5
wt, _ = M.prepare_weights(L, st, wgt)
hi, lo = M.split_weights_f16(L, st, wt)
act = M.split_activation_f16(L, st, x) if not hasattr(M, "split_activation_mod") else None
act = M.split_activation_f16(L, st, x)
for tune, name in [(16, "tiled"), (0, "rows"),  (32, "rows, no DMA"), (64, "rows, no epilogue"), (96, "rows, no DMA no epilogue"), (128, "rows, no MFMA loop"), (224, "rows, nothing")]:
    L.hf_debug_set_tuning(tune)
    t = timeit(lambda: M.modconv3x3_f16_pre(L, st, act, hi, lo, 3, dm, nz, nw, bias, rgb=(rgb_w, rgb_s), want_out=False))
    print(f"{name:28s} {t:8.1f} us", flush=True)
L.hf_debug_set_tuning(0)
