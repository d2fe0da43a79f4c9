"""Per-layer timing of the generator's fast path at batch 8 (pre-split inputs): same-resolution convs (fp32 + split
output / fused ToRGB), fused and two-pass upsampling convs - under the stage-DMA issue schedules of
hf_debug_set_tuning (0 = library rule, 1 = early, 2 = spread)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hairfastgan_amd import _marshal as M
from hairfastgan_amd._runtime import lib, stream
from oracle import ref_stylegan2 as O

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

dev = torch.device("cuda:0"); L, st = lib(), stream()
k4 = O.blur_kernel_1d_to_2d(gain=4.0).to(dev); fac = M.blur_factors(k4)
B = int(os.environ.get("PROBE_BATCH", "8"))
modes = [int(m) for m in os.environ.get("PROBE_TUNE", "0,1,2").split(",")]
print("layer".ljust(28) + " | " + " | ".join(f"tune {m}: us  TF/s" for m in modes))

def report(name, flops, fn):
    cells = []
    for m in modes:
        L.hf_debug_set_tuning(m)
        t = timeit(fn)
        cells.append(f"{t:8.1f} {flops / t * 1e-6:6.1f}")
    L.hf_debug_set_tuning(0)
    print(name.ljust(28) + " | " + " | ".join(cells), flush=True)

for c, h, rgb in [(512, 64, False), (256, 128, False), (128, 256, False), (64, 512, True), (32, 1024, True)]:
    torch.manual_seed(0)
    x = torch.randn(B, c, h, h, device=dev)
    wgt = torch.randn(1, c, c, 3, 3, device=dev)
    s, d, s2 = (torch.rand(B, c, device=dev) + 0.5 for _ in range(3))
    nz, nw, bias = torch.randn(B, 1, h, h, device=dev), torch.tensor([0.3], device=dev), torch.randn(c, device=dev)
    wt, _ = M.prepare_weights(L, st, wgt)
    hi, lo = M.split_weights_f16(L, st, wt)
    xs = M.SplitActivation(*M.split_activation_reference(x, s), None)
    del x
    rgbp = (torch.randn(c, 3, device=dev), torch.rand(B, c, device=dev) + 0.5) if rgb else None
    last = h == 1024
    fn = lambda: M.modconv3x3_f16_pre(L, st, xs, hi, lo, 3, d, nz, nw, bias, rgb=rgbp, want_out=not rgb,
                                      split_for=None if last else s2)
    report(f"same {c:3d}->{c:3d} @{h:4d}" + (" rgb" if rgb else ""), 2.0 * c * c * 9 * h * h * B, fn)

for cin, cout, h in [(64, 32, 512), (128, 64, 256), (256, 128, 128), (512, 256, 64), (512, 512, 32)]:
    torch.manual_seed(0)
    x = torch.randn(B, cin, h, h, device=dev)
    wgt = torch.randn(1, cout, cin, 3, 3, device=dev)
    s, d = torch.rand(B, cin, device=dev) + 0.5, torch.rand(B, cout, device=dev) + 0.5
    s2 = torch.rand(B, cout, device=dev) + 0.5
    nz, nw, bias = torch.randn(B, 1, 2 * h, 2 * h, device=dev), torch.tensor([0.3], device=dev), torch.randn(cout, device=dev)
    wt, _ = M.prepare_weights(L, st, wgt)
    hi, lo = M.split_weights_f16(L, st, wt)
    xs = M.SplitActivation(*M.split_activation_reference(x, s), None)
    del x
    fl = 2.0 * cin * cout * 9 * h * h * B
    report(f"up2p {cin:3d}->{cout:3d} {h:4d}->{2*h:4d}", fl,
           lambda: M.modconv3x3_up(L, st, xs, wt, None, d, k4, nz, nw, bias, f16=(hi, lo, 3), split_for=(None, s2, True)))
    if M.modconv3x3_up_fused_supported(cin, cout, h, h):
        report(f"upfu {cin:3d}->{cout:3d} {h:4d}->{2*h:4d}", fl,
               lambda: M.modconv3x3_up_fused(L, st, xs, hi, lo, None, d, fac, nz, nw, bias, split_for=s2))
