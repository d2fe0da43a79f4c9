"""Timeline of block 0 (HF_H_TRACE build of csrc/convh.hip): python trace_layer.py cin cout res [batch]"""
import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hairfastgan_amd import _marshal as M
from hairfastgan_amd._runtime import lib, stream
cin, cout, r = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 8
L = lib(); st = stream(); dev = torch.device("cuda:0")
x = torch.randn(B, cin, r, r, device=dev)
wt, wsq = M.prepare_weights(L, st, torch.randn(1, cout, cin, 3, 3, device=dev))
hi, lo = M.split_weights_f16(L, st, wt)
s = torch.rand(B, cin, device=dev) + 0.5; d = torch.rand(B, cout, device=dev) + 0.5
nz = torch.randn(1, 1, r, r, device=dev); nw = torch.tensor([0.1], device=dev); bias = torch.randn(cout, device=dev)
for _ in range(2):
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
