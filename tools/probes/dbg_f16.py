import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hairfastgan_amd import _marshal as M
from hairfastgan_amd._runtime import lib, stream
L = lib(); st = stream(); dev = torch.device("cuda:0")
torch.set_printoptions(linewidth=250, precision=10, sci_mode=False)
(B, cin, cout, H, W) = (1, 48, 64, 8, 32)
torch.manual_seed(7)
x = torch.randn(B, cin, H, W, device=dev)
wgt = torch.randn(1, cout, cin, 3, 3, device=dev)
s = torch.rand(B, cin, device=dev) + 0.5
print("x", x[0, 34, 7, 3].item(), "s", s[0, 34].item(), "v", (x[0, 34, 7, 3] * s[0, 34]).item())
v = x[0, 34, 7, 3] * s[0, 34]
h = v.half(); l = (v - h.float()).half()
print("hi", h.item(), "lo", l.item(), "resid", (v.double() - h.double() - l.double()).item())
print("neighbours", (x[0, 34, 6:8, 2:5] * s[0, 34]).tolist())
print("absmax x*s over tensor", float((x * s[:, :, None, None]).abs().max()))
