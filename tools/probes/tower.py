"""Per-layer timing of the generator's 4^2-32^2 tower (512 -> 512 channels): the fp32 split-K kernels against the
fp16-core kernels where the latter take the shape."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hairfastgan_amd import _marshal as M
from hairfastgan_amd._runtime import lib, stream
from oracle import ref_stylegan2 as O

def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

dev = torch.device("cuda:0"); L, st = lib(), stream()
k4 = O.blur_kernel_1d_to_2d(gain=4.0).to(dev)
c = 512
for B in (8, 1, 3):
    print(f"--- batch {B}")
    for h in (4, 8, 16, 32):
        torch.manual_seed(0)
        x = torch.randn(B, c, h, h, device=dev)
        wgt = torch.randn(1, c, c, 3, 3, device=dev)
        s, d = torch.rand(B, c, device=dev) + 0.5, torch.rand(B, c, device=dev) + 0.5
        nz, nw, bias = torch.randn(B, 1, h, h, device=dev), torch.tensor([0.3], device=dev), torch.randn(c, device=dev)
        nz2 = torch.randn(B, 1, 2 * h, 2 * h, device=dev)
        wt, _ = M.prepare_weights(L, st, wgt)
        hi, lo = M.split_weights_f16(L, st, wt)
        fl = 2.0 * c * c * 9 * h * h * B
        t32 = timeit(lambda: M.modconv3x3(L, st, x, wt, s, d, nz, nw, bias))
        line = f"same {h:3d}^2: fp32 {t32:7.1f} us ({fl / t32 * 1e-6:6.1f} TF/s)"
        try:
            y = M.modconv3x3_f16(L, st, x, hi, lo, 3, s, d, nz, nw, bias)
            ref = M.modconv3x3(L, st, x, wt, s, d, nz, nw, bias)
            err = float((y - ref).abs().max() / ref.abs().max())
            t16 = timeit(lambda: M.modconv3x3_f16(L, st, x, hi, lo, 3, s, d, nz, nw, bias))
            line += f" | f16x3 {t16:7.1f} us ({fl / t16 * 1e-6:6.1f} TF/s, rel err {err:.1e}, path {L.hf_debug_last_path()})"
        except Exception as e:
            line += f" | f16x3 n/a"
        if h <= 32:
            w9 = M.split_weights_small(L, st, wt)
            ys = M.modconv3x3_small(L, st, x, w9, 3, s, d, nz, nw, bias, c)
            ref = M.modconv3x3(L, st, x, wt, s, d, nz, nw, bias)
            err = float((ys - ref).abs().max() / ref.abs().max())
            ts = timeit(lambda: M.modconv3x3_small(L, st, x, w9, 3, s, d, nz, nw, bias, c))
            line += f" | tap-GEMM {ts:7.1f} us (rel err {err:.1e})"
        print(line, flush=True)
        if h <= 16:
            tu = timeit(lambda: M.modconv3x3_up(L, st, x, wt, s, d, k4, nz2, nw, bias))
            line = f"up   {h:3d}->{2*h:3d}: fp32 {tu:7.1f} us"
            if M.modconv3x3_up_f16_supported(c, c, h, h):
                tu16 = timeit(lambda: M.modconv3x3_up(L, st, x, wt, s, d, k4, nz2, nw, bias, f16=(hi, lo, 3)))
                line += f" | f16x3 {tu16:7.1f} us"
            def up_small():
                tmp = M.modconv3x3_small(L, st, x, w9, 3, s, d, None, None, None, c, upsample=True)
                out = tmp.new_empty((B, c, 2 * h, 2 * h))
                from hairfastgan_amd._lib import check
                check(L, L.hf_blur_noise_bias_act_f32(out.data_ptr(), tmp.data_ptr(), k4.data_ptr(), nz2.data_ptr(), nw.data_ptr(),
                                                      4 * h * h, bias.data_ptr(), B, c, 2 * h + 1, 2 * h + 1, tmp.shape[3], 0.2, 2 ** 0.5, st), "blur")
                return out
            ref = M.modconv3x3_up(L, st, x, wt, s, d, k4, nz2, nw, bias)
            err = float((up_small() - ref).abs().max() / ref.abs().max())
            tus = timeit(up_small)
            line += f" | tap-GEMM+blur {tus:7.1f} us (rel err {err:.1e})"
            print(line, flush=True)
