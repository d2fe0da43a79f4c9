"""Launch one f16x3 conv layer a few times (for rocprofv3 --pmc passes):
python one_layer.py cin cout res [batch] [same|pre|up|uppre]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hairfastgan_amd import _marshal as M
from hairfastgan_amd._runtime import lib, stream
cin, cout, r = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 8
up = len(sys.argv) > 5 and sys.argv[5].startswith("up")
pre = len(sys.argv) > 5 and sys.argv[5].endswith("pre")
L = lib(); st = stream(); dev = torch.device("cuda:0")
x = torch.randn(B, cin, r, r, device=dev)
wt, wsq = M.prepare_weights(L, st, torch.randn(1, cout, cin, 3, 3, device=dev))
hi, lo = M.split_weights_f16(L, st, wt)
s = torch.rand(B, cin, device=dev) + 0.5; d = torch.rand(B, cout, device=dev) + 0.5
oh = 2 * r if up else r
nz = torch.randn(1, 1, oh, oh, device=dev); nw = torch.tensor([0.1], device=dev); bias = torch.randn(cout, device=dev)
k4 = torch.tensor([1., 3., 3., 1.], device=dev); k4 = k4[None] * k4[:, None] / 16
xin = M.SplitActivation(*M.split_activation_reference(x, s), None) if pre else x
for _ in range(3):
    if up:
        y = M.modconv3x3_up(L, st, xin, wt, None if pre else s, d, k4, nz, nw, bias, f16=(hi, lo, 3))
    elif pre:
        y = M.modconv3x3_f16_pre(L, st, xin, hi, lo, 3, d, nz, nw, bias)
    else:
        y = M.modconv3x3_f16(L, st, x, hi, lo, 3, s, d, nz, nw, bias)
torch.cuda.synchronize()
