"""Timeline of block 0 (HF_H_TRACE build of csrc/convh.hip): python trace_layer.py cin cout res [batch]"""
import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hairfastgan_amd import _marshal as M
from hairfastgan_amd._runtime import lib, stream
cin, cout, r = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 8
up = len(sys.argv) > 5 and sys.argv[5].startswith('up')
pre = len(sys.argv) > 5 and sys.argv[5].endswith('pre')
L = lib(); st = stream(); dev = torch.device("cuda:0")
x = torch.randn(B, cin, r, r, device=dev)
wt, wsq = M.prepare_weights(L, st, torch.randn(1, cout, cin, 3, 3, device=dev))
hi, lo = M.split_weights_f16(L, st, wt)
s = torch.rand(B, cin, device=dev) + 0.5; d = torch.rand(B, cout, device=dev) + 0.5
oh = 2 * r if up else r
nz = torch.randn(1, 1, oh, oh, device=dev)
k4 = torch.tensor([1., 3., 3., 1.], device=dev); k4 = k4[None] * k4[:, None] / 16; nw = torch.tensor([0.1], device=dev); bias = torch.randn(cout, device=dev)
for _ in range(2):
    xin = M.SplitActivation(*M.split_activation_reference(x, s), None) if pre else x
    if up:
        y = M.modconv3x3_up(L, st, xin, wt, None if pre else s, d, k4, nz, nw, bias, f16=(hi, lo, 3))
    elif pre:
        y = M.modconv3x3_f16_pre(L, st, xin, hi, lo, 3, d, nz, nw, bias)
    else:
        y = M.modconv3x3_f16(L, st, x, hi, lo, 3, s, d, nz, nw, bias)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 512))()
L.hf_debug_read_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
print("rc", L.hf_debug_read_trace(buf, 8 * 512))
names = {1: "tile", 2: "mfma_done", 3: "barrier_done", 4: "epilogue_issued", 5: "pre_convert", 6: "post_convert"}
for wave in (0, 5):
    ev = [(buf[wave * 512 + i] >> 56, buf[wave * 512 + i] & ((1 << 56) - 1)) for i in range(512)]
    ev = [(i, t) for i, t in ev if i]
    t0 = ev[0][1]
    print("wave", wave, "events", len(ev))
    prev = t0
    for i, t in ev[20:75]:
        print(f"  {names.get(i, i):16s} t={t - t0:8d}  +{t - prev:6d}")
        prev = t
