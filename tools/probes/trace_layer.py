"""Timeline of block 0 (HF_H_TRACE build of csrc/convh.hip, loaded through HAIRFAST_HIP_LIB):
python trace_layer.py cin cout res [batch] [mode]   mode: '' | pre | presplit (split output only) | prergb (fused ToRGB, no fp32 output) | up | uppre | fuse"""
import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hairfastgan_amd import _marshal as M
from hairfastgan_amd._runtime import lib, stream
cin, cout, r = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 8
mode = sys.argv[5] if len(sys.argv) > 5 else ''
up = mode.startswith('up')
pre = mode.endswith('pre') or mode in ('prergb', 'prergbsplit', 'fuse', 'presplit')
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
    if mode == 'fuse':
        s2 = torch.rand(B, cout, device=dev) + 0.5
        y = M.modconv3x3_up_fused(L, st, xin, hi, lo, None, d, M.blur_factors(k4 * 4), torch.randn(B, 1, 2 * r, 2 * r, device=dev), nw, bias, split_for=s2)
    elif mode == 'prergb':
        rgbp = (torch.randn(cout, 3, device=dev), torch.rand(B, cout, device=dev) + 0.5)
        y = M.modconv3x3_f16_pre(L, st, xin, hi, lo, 3, d, nz, nw, bias, rgb=rgbp, want_out=False)
    elif mode == 'prergbsplit':  # the generator's 64^2 .. 512^2 same-resolution layers: ToRGB slabs + split output, no fp32 activation
        rgbp = (torch.randn(cout, 3, device=dev), torch.rand(B, cout, device=dev) + 0.5)
        s2 = torch.rand(B, cout, device=dev) + 0.5
        y = M.modconv3x3_f16_pre(L, st, xin, hi, lo, 3, d, nz, nw, bias, rgb=rgbp, want_out=False, split_for=s2)
    elif mode == 'presplit':
        s2 = torch.rand(B, cout, device=dev) + 0.5
        y = M.modconv3x3_f16_pre(L, st, xin, hi, lo, 3, d, nz, nw, bias, want_out=False, split_for=s2)
    elif up:
        y = M.modconv3x3_up(L, st, xin, wt, None if pre else s, d, k4, nz, nw, bias, f16=(hi, lo, 3))
    elif pre:
        y = M.modconv3x3_f16_pre(L, st, xin, hi, lo, 3, d, nz, nw, bias)
    else:
        y = M.modconv3x3_f16(L, st, x, hi, lo, 3, s, d, nz, nw, bias)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 512))()
L.hf_debug_read_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
print("rc", L.hf_debug_read_trace(buf, 8 * 512))
names = {1: "tile", 2: "mfma_done", 3: "barrier_done", 4: "epilogue_issued", 5: "pre_convert", 6: "post_convert", 7: "geom_done", 8: "next_located"}
import collections
for wave in (0, 5):
    ev = [(buf[wave * 512 + i] >> 56, buf[wave * 512 + i] & ((1 << 56) - 1)) for i in range(512)]
    ev = [(i, t) for i, t in ev if i]
    print("wave", wave, "events", len(ev))
    # duration from each event to the next one, grouped by (event id -> next id)
    seg = collections.defaultdict(list)
    for (a, ta), (b, tb) in zip(ev[30:], ev[31:]):
        seg[(a, b)].append(tb - ta)
    nm = lambda i: names.get(i, f"step{i-10}" if 10 <= i < 20 else f"mfma{i-20}" if 20 <= i < 30 else str(i))
    for (a, b), v in sorted(seg.items()):
        v.sort()
        print(f"  {nm(a):>16s} -> {nm(b):16s} n={len(v):3d}  median {v[len(v)//2]:6d}  min {v[0]:6d}  max {v[-1]:6d}")
    tiles = [t for i, t in ev if i == 1]
    if len(tiles) > 3:
        per = [(b - a) for a, b in zip(tiles[1:-1], tiles[2:])]
        print(f"  tiles {len(tiles)}: ticks per tile min {min(per)} median {sorted(per)[len(per)//2]} max {max(per)}")
