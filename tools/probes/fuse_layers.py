"""Per-layer timing of the upsampling StyledConv: two-pass (transposed conv + blur pass) vs the fused kernel
(8-row / 4-wave and 16-row / 8-wave forms), pre-split input, split output - the generator's fast path."""
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
print(f"{'layer':22s} | two-pass us | fused (split out) us | fused (fp32 out) us")
for B, cin, cout, h in [(8, 64, 32, 512), (8, 128, 64, 256), (8, 256, 128, 128), (8, 512, 256, 64), (1, 64, 32, 512), (1, 128, 64, 256), (2, 64, 32, 512)]:
    torch.manual_seed(0)
    x = torch.randn(B, cin, h, h, device=dev)
    wgt = torch.randn(1, cout, cin, 3, 3, device=dev)
    s, d = torch.rand(B, cin, device=dev) + 0.5, torch.rand(B, cout, device=dev) + 0.5
    s2 = torch.rand(B, cout, device=dev) + 0.5
    nz, nw, bias = torch.randn(B, 1, 2 * h, 2 * h, device=dev), torch.tensor([0.3], device=dev), torch.randn(cout, device=dev)
    wt, _ = M.prepare_weights(L, st, wgt)
    hi, lo = M.split_weights_f16(L, st, wt)
    xs = M.SplitActivation(*M.split_activation_reference(x, s), None)
    t2 = timeit(lambda: M.modconv3x3_up(L, st, xs, wt, None, d, k4, nz, nw, bias, f16=(hi, lo, 3), split_for=(None, s2, True)))
    tf = timeit(lambda: M.modconv3x3_up_fused(L, st, xs, hi, lo, None, d, fac, nz, nw, bias, split_for=s2))
    tf32 = timeit(lambda: M.modconv3x3_up_fused(L, st, xs, hi, lo, None, d, fac, nz, nw, bias))
    print(f"B{B} {cin:3d}->{cout:3d} {h:4d}->{2*h:4d} | {t2:10.1f} | {tf:10.1f} | {tf32:10.1f}", flush=True)
