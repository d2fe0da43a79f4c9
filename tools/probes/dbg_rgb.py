"""Debug: fused ToRGB raw product of hf_modconv3x3_f16_rgb_f32 against a torch contraction of the conv output."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hairfastgan_amd import _marshal as M
from hairfastgan_amd._runtime import lib as _lib_fn, stream
dev = torch.device("cuda:0"); lib, st = _lib_fn(), stream()
for shape in [(2, 64, 64, 64, 64), (1, 32, 32, 96, 128), (2, 32, 64, 16, 32)]:
    B, cin, cout, H, W = shape
    torch.manual_seed(9)
    r = lambda *sz: torch.randn(*sz, device=dev)
    x, wgt = r(B, cin, H, W), r(1, cout, cin, 3, 3)
    mw, mb, sty = r(cin, 16), r(cin), r(B, 16)
    nz, nw, bias = r(B, 1, H, W), torch.tensor([0.3], device=dev), r(cout)
    wrgb, mwr, mbr, styr = r(1, 3, cout, 1, 1), r(cout, 16), r(cout), r(B, 16)
    wt, wsq = M.prepare_weights(lib, st, wgt)
    s = M.modulation(lib, st, sty, mw, mb)
    dm = M.demod(lib, st, s, wsq)
    hi, lo = M.split_weights_f16(lib, st, wt)
    wtr, _ = M.prepare_weights(lib, st, wrgb)
    sr = M.modulation(lib, st, styr, mwr, mbr)
    for blocks in (0, 3):
        lib.hf_debug_set_persistent_blocks(blocks)
        y, raw = M.modconv3x3_f16(lib, st, x, hi, lo, 3, s, dm, nz, nw, bias, rgb=(wtr, sr))
        torch.cuda.synchronize()
        w2 = wtr.reshape(cout, 3)  # prepared 1x1 weights [tap=1][ci][co=3]
        ref = torch.einsum("bkhw,kc,bk->bchw", y.double(), w2.double(), sr.double()).float()
        d = (raw - ref).abs()
        bad = (d > 1e-3 * ref.abs().max()).nonzero()
        print(shape, "blocks", blocks, "path", lib.hf_debug_last_path(), "max diff", float(d.max()), "ref max", float(ref.abs().max()), "n bad", len(bad))
        if len(bad):
            print("  first bad", bad[:8].tolist(), "rows", sorted(set(bad[:, 2].tolist()))[:20], "cols", sorted(set(bad[:, 3].tolist()))[:40])
    lib.hf_debug_set_persistent_blocks(0)
