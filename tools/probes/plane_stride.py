"""Same 32->32 (and 64->64) f16x3 conv, same pixel count, different plane sizes: is the
high-resolution slowdown tied to the 4 MB channel-plane stride (TLB reach / DRAM locality)?"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hairfastgan_amd import _marshal as M
from hairfastgan_amd._runtime import lib, stream
L = lib(); st = stream(); dev = torch.device("cuda:0")
def run(B, c, r, plain=False):
    x = torch.randn(B, c, r, r, device=dev)
    wt, _ = M.prepare_weights(L, st, torch.randn(1, c, c, 3, 3, device=dev))
    hi, lo = M.split_weights_f16(L, st, wt)
    s = torch.rand(B, c, device=dev) + 0.5; d = torch.rand(B, c, device=dev) + 0.5
    nz = torch.randn(1, 1, r, r, device=dev); nw = torch.tensor([0.1], device=dev); bias = torch.randn(c, device=dev)
    out = torch.empty(B, c, r, r, device=dev)
    fn = lambda: L.hf_modconv3x3_f16_f32(out.data_ptr(), x.data_ptr(), hi.data_ptr(), lo.data_ptr(), 3, s.data_ptr(), None if plain else d.data_ptr(),
                                         None if plain else nz.data_ptr(), None if plain else nw.data_ptr(), 0, None if plain else bias.data_ptr(), B, c, c, r, r, 0.2, 1.414, st)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5 * 1e-3
    fl = 2.0 * c * c * 9 * r * r * B
    print(f"B={B:5d} c={c} res={r:5d} plane={r*r*4/1024:8.0f} KiB : {t*1e3:7.3f} ms  {fl/t/1e12:6.1f} TF  {(2*B*c*r*r*4)/t/1e9:6.0f} GB/s")
for r in (1024, 992, 1000, 1056, 960):
    run(8, 32, r)
for r in (512, 496, 544):
    run(8, 64, r)
