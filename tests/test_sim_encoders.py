"""CPU tests of the encoder kernels and module mirrors (kernel sources interpreted by
tests/hipsim) against the oracle and the golden vectors produced by the real reference."""
import sys

import pytest
import torch
import torch.nn.functional as F

from hairfastgan_amd import _marshal as M
from oracle import cases as C
from oracle import ref_encoders as E

TOL = 2e-5


def maxdiff(a, b):
    return float((a.double() - b.double()).abs().max())


@pytest.fixture()
def sim_backend(simlib, monkeypatch):
    import hairfastgan_amd.encoders  # noqa: F401

    mods = [sys.modules[n] for n in ("hairfastgan_amd.encoders._fused", "hairfastgan_amd.encoders.e4e",
                                     "hairfastgan_amd.encoders.fs_encoder", "hairfastgan_amd.encoders.post_process")]
    for mod in mods:
        monkeypatch.setattr(mod, "lib", lambda: simlib)
        monkeypatch.setattr(mod, "stream", lambda: None)
        monkeypatch.setattr(mod, "require_gpu", lambda *a: None)
    return mods


@pytest.mark.parametrize("k,stride,B,cin,cout,H,W", [(3, 2, 2, 16, 24, 9, 13), (1, 2, 1, 12, 40, 10, 7), (1, 1, 3, 8, 8, 5, 5),
                                                      (3, 1, 1, 3, 64, 20, 36), (3, 2, 1, 8, 8, 1, 1), (3, 1, 2, 16, 64, 16, 32)])
def test_conv2d_variants(simlib, k, stride, B, cin, cout, H, W):
    torch.manual_seed(k * 10 + stride + cin)
    x = torch.randn(B, cin, H, W)
    w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
    a, t = torch.rand(cin) + 0.5, torch.randn(cin) * 0.2
    g, bsh, slope = torch.rand(cout) + 0.5, torch.randn(cout) * 0.2, torch.rand(cout) * 0.5
    wt = M.conv_prepare(simlib, None, w)
    ref_core = F.conv2d(x * a.view(1, -1, 1, 1) + t.view(1, -1, 1, 1), w, stride=stride, padding=k // 2)
    ref = F.prelu(ref_core * g.view(1, -1, 1, 1) + bsh.view(1, -1, 1, 1), slope)
    res = torch.randn_like(ref)
    y = M.conv2d(simlib, None, x, wt, k, stride, in_scale=a, in_shift=t, out_scale=g, bias=bsh, act=M.ACT_PRELU,
                 slope=slope, residual=res)
    assert maxdiff(y, ref + res) < TOL * max(1.0, float(ref.abs().max()))
    y = M.conv2d(simlib, None, x, wt, k, stride, bias=bsh, act=M.ACT_LRELU, alpha=0.01)
    ref = F.leaky_relu(F.conv2d(x, w, bsh, stride=stride, padding=k // 2), 0.01)
    assert maxdiff(y, ref) < TOL * max(1.0, float(ref.abs().max()))


def test_small_ops(simlib):
    torch.manual_seed(0)
    x, y = torch.randn(2, 3, 5, 7), torch.randn(2, 3, 10, 14)
    assert maxdiff(M.upsample_bilinear_add(simlib, None, x, y),
                   F.interpolate(x, size=(10, 14), mode="bilinear", align_corners=True) + y) < 1e-6
    x = torch.randn(2, 5, 16, 8)
    out = torch.zeros(2, 9, 3, 3)
    M.adaptive_avgpool_into(simlib, None, out, x, 2)
    assert maxdiff(out[:, 2:7], F.adaptive_avg_pool2d(x, (3, 3))) < 1e-6 and float(out[:, :2].abs().max()) == 0
    x = torch.randn(2, 3, 8, 12)
    assert maxdiff(M.downscale2x(simlib, None, x), F.interpolate(x, scale_factor=0.5, mode="bilinear")) < 1e-6
    x, w, b = torch.randn(3, 40), torch.randn(7, 40), torch.randn(7)
    assert maxdiff(M.linear(simlib, None, x, w, b, 0.5), F.linear(x, w * 0.5, b)) < 1e-5
    x, w = torch.randn(2, 33), torch.randn(5, 33)  # unaligned K -> scalar path
    assert maxdiff(M.linear(simlib, None, x, w, None, 1.0), F.linear(x, w)) < 1e-5
    x = torch.randn(2, 4, 6, 6)
    pm = M.plane_mean(simlib, None, x)
    assert maxdiff(pm, x.mean((2, 3))) < 1e-6
    fc1, fc2 = torch.randn(2, 4), torch.randn(4, 2)
    gate = M.se_gate(simlib, None, pm, fc1, fc2)
    assert maxdiff(gate, torch.sigmoid(F.linear(F.relu(F.linear(pm, fc1)), fc2))) < 1e-6
    sc = torch.randn(2, 4, 12, 12)
    assert maxdiff(M.scale_shortcut_add(simlib, None, x, gate, sc, 2), x * gate[:, :, None, None] + sc[:, :, ::2, ::2]) < 1e-6
    g, be, mu, var, cb = torch.rand(6) + 0.5, torch.randn(6), torch.randn(6), torch.rand(6) + 0.5, torch.randn(6)
    s, t = M.bn_fold(simlib, None, g, be, mu, var, 1e-5, cb)
    assert maxdiff(s, g / torch.sqrt(var + 1e-5)) < 1e-6 and maxdiff(t, be + (cb - mu) * g / torch.sqrt(var + 1e-5)) < 1e-5
    a, bvec = torch.randn(3, 10), torch.randn(10)
    assert maxdiff(M.add_bcast(simlib, None, a, bvec), a + bvec) == 0


def _load(mod, P):
    sd = {k[2:]: v for k, v in P.items()}
    for k in mod.state_dict():
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.long)
    mod.load_state_dict(sd)
    return mod.eval()


@pytest.mark.parametrize("name", list(C.IRSE_UNIT_CASES))
def test_irse_unit(sim_backend, golden, name):
    from hairfastgan_amd.encoders.e4e import bottleneck_IR_SE

    in_c, depth, stride, B, H, W = C.IRSE_UNIT_CASES[name]
    m = _load(bottleneck_IR_SE(in_c, depth, stride), C.params_from_shapes(name, C.irse_unit_shapes(in_c, depth)))
    y = m(C.unit_input(name, (B, in_c, H, W)))
    assert maxdiff(y, torch.from_numpy(golden("encoder_units.npz")[name])) < TOL


@pytest.mark.parametrize("name", list(C.IBASIC_CASES))
def test_ibasic_block(sim_backend, golden, name):
    from hairfastgan_amd.encoders.fs_encoder import IBasicBlock
    from torch import nn

    in_c, planes, stride, B, H, W = C.IBASIC_CASES[name]
    ds = None
    if stride != 1 or in_c != planes:
        ds = nn.Sequential(nn.Conv2d(in_c, planes, 1, stride, bias=False), nn.BatchNorm2d(planes, eps=1e-05))
    m = _load(IBasicBlock(in_c, planes, stride, ds), C.params_from_shapes(name, C.ibasic_shapes(in_c, planes, stride)))
    y = m(C.unit_input(name, (B, in_c, H, W)))
    assert maxdiff(y, torch.from_numpy(golden("encoder_units.npz")[name])) < TOL


@pytest.mark.parametrize("name", list(C.STYLE_BLOCK_CASES))
def test_gradual_style_block(sim_backend, golden, name):
    from hairfastgan_amd.encoders.e4e import GradualStyleBlock

    c, spatial, B = C.STYLE_BLOCK_CASES[name]
    m = _load(GradualStyleBlock(c, c, spatial), C.params_from_shapes(name, C.style_block_shapes(c, spatial)))
    y = m(C.unit_input(name, (B, c, spatial, spatial)))
    assert maxdiff(y, torch.from_numpy(golden("encoder_units.npz")[name])) < TOL


def test_state_dicts_match_reference_layout():
    import argparse

    from hairfastgan_amd.encoders import Encoder4Editing, fs_encoder_v2

    e = Encoder4Editing(50, "ir_se", argparse.Namespace(stylegan_size=1024))
    assert {k: tuple(v.shape) for k, v in e.state_dict().items()} == E.e4e_param_shapes()
    assert list(e.state_dict()) == list(E.e4e_param_shapes())
    f = fs_encoder_v2(stride=(2, 2))
    assert {k: tuple(v.shape) for k, v in f.state_dict().items()} == E.fs_param_shapes()
    assert list(f.state_dict()) == list(E.fs_param_shapes())


def test_grouped_conv(simlib):
    """Grouped launches (the e4e style heads): shared and per-group inputs, split-K and direct."""
    torch.manual_seed(11)
    G, B, cin, cout = 3, 2, 16, 24
    for (H, W, shared) in [(8, 8, True), (9, 6, False)]:
        w = torch.randn(G, cout, cin, 3, 3) / (cin * 9) ** 0.5
        bias = torch.randn(G, cout) * 0.2
        x = torch.randn(B, cin, H, W) if shared else torch.randn(G, B, cin, H, W)
        wt = torch.stack([M.conv_prepare(simlib, None, w[g]) for g in range(G)])
        y = M.conv2d(simlib, None, x, wt, 3, 2, bias=bias, act=M.ACT_LRELU, alpha=0.01, groups=G, x_shared=shared)
        for g in range(G):
            xin = x if shared else x[g]
            ref = F.leaky_relu(F.conv2d(xin, w[g], bias[g], stride=2, padding=1), 0.01)
            assert maxdiff(y[g], ref) < TOL * max(1.0, float(ref.abs().max())), (H, W, shared, g)


@pytest.mark.parametrize("B,cin,cout,H,W,code", [(4, 8, 64, 128, 128, 245), (2, 8, 128, 256, 256, 244)])
def test_pipelined_stride2(simlib, B, cin, cout, H, W, code):
    """The double-buffered kernel in its stride-2 form (large strided 3x3 convs of the encoders)."""
    torch.manual_seed(B)
    x = torch.randn(B, cin, H, W)
    w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    g, bsh = torch.rand(cout) + 0.5, torch.randn(cout) * 0.2
    y = M.conv2d(simlib, None, x, M.conv_prepare(simlib, None, w), 3, 2, out_scale=g, bias=bsh)
    assert simlib.hf_debug_last_path() == code
    ref = F.conv2d(x, w, stride=2, padding=1) * g.view(1, -1, 1, 1) + bsh.view(1, -1, 1, 1)
    assert maxdiff(y, ref) < TOL * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("nterms,tol", [(3, 5e-6), (1, 4e-3)])
@pytest.mark.parametrize("stride,B,cin,cout,H,W,path", [
    (1, 2, 32, 64, 16, 32, 603), (1, 1, 16, 128, 20, 40, 603), (1, 2, 48, 64, 16, 16, 603),   # tw 32 / ragged / tw 16
    (2, 2, 32, 64, 16, 64, 602), (2, 1, 16, 64, 31, 33, 602), (2, 2, 32, 64, 32, 32, 602),   # stride 2: tw 32 / odd planes / tw 16
])
def test_conv2d_f16_matrix_cores(simlib, nterms, tol, stride, B, cin, cout, H, W, path):
    """csrc/convh_enc.hip: the encoders' 3x3 convs on the fp16 matrix cores (stride 1 and the
    parity-split stride 2 form), pre-conv affine on real pixels only, post-conv affine, PReLU,
    residual - against the fp32-MFMA kernel and torch."""
    torch.manual_seed(stride * 100 + cin + W)
    x = torch.randn(B, cin, H, W)
    w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    a, t = torch.rand(cin) + 0.5, torch.randn(cin) * 0.2
    g, bsh, slope = torch.rand(cout) + 0.5, torch.randn(cout) * 0.2, torch.rand(cout) * 0.5
    wt = M.conv_prepare(simlib, None, w)
    assert M.conv2d_f16_supported(cin, cout, H, W, 3, stride)
    hi, lo = M.conv_split_weights_f16(simlib, None, wt)
    ref_core = F.conv2d(x * a.view(1, -1, 1, 1) + t.view(1, -1, 1, 1), w, stride=stride, padding=1)
    ref = F.prelu(ref_core * g.view(1, -1, 1, 1) + bsh.view(1, -1, 1, 1), slope)
    res = torch.randn_like(ref)
    kw = dict(in_scale=a, in_shift=t, out_scale=g, bias=bsh, act=M.ACT_PRELU, slope=slope, residual=res)
    y32 = M.conv2d(simlib, None, x, wt, 3, stride, **kw)
    y = M.conv2d_f16(simlib, None, x, hi, lo, nterms, cout, stride, **kw)
    assert simlib.hf_debug_last_path() == path
    scale = max(1.0, float(ref.abs().max()))
    assert maxdiff(y, y32) < tol * scale
    assert maxdiff(y, ref + res) < max(tol, TOL) * scale
    # plain conv + bias + LeakyReLU (the style heads' form)
    y = M.conv2d_f16(simlib, None, x, hi, lo, nterms, cout, stride, bias=bsh, act=M.ACT_LRELU, alpha=0.01)
    ref = F.leaky_relu(F.conv2d(x, w, bsh, stride=stride, padding=1), 0.01)
    assert maxdiff(y, ref) < max(tol, TOL) * max(1.0, float(ref.abs().max()))


def test_conv2d_f16_grouped(simlib):
    """Grouped launch (the e4e style heads of a family): per-group weight blobs with their own
    pre-scale trailers, shared and per-group inputs."""
    torch.manual_seed(9)
    G, B, cin, cout, H, W = 3, 2, 16, 64, 32, 32
    x = torch.randn(B, cin, H, W)
    ws = [torch.randn(cout, cin, 3, 3) * (0.02 * 10 ** gi) for gi in range(G)]  # very different magnitudes per group
    bias = torch.randn(G, cout)
    wt = torch.stack([M.conv_prepare(simlib, None, w) for w in ws]).contiguous()
    hi, lo = M.conv_split_weights_f16(simlib, None, wt)
    y = M.conv2d_f16(simlib, None, x, hi, lo, 3, cout, 2, bias=bias, act=M.ACT_LRELU, alpha=0.01, groups=G, x_shared=True)
    assert y.shape == (G, B, cout, 16, 16)
    for gi in range(G):
        ref = F.leaky_relu(F.conv2d(x, ws[gi], bias[gi], stride=2, padding=1), 0.01)
        assert maxdiff(y[gi], ref) < TOL * max(1.0, float(ref.abs().max()))
    x2 = torch.randn(G, B, cin, H, W)
    y2 = M.conv2d_f16(simlib, None, x2, hi, lo, 3, cout, 1, bias=bias, groups=G, x_shared=False)
    for gi in range(G):
        ref = F.conv2d(x2[gi], ws[gi], bias[gi], padding=1)
        assert maxdiff(y2[gi], ref) < TOL * max(1.0, float(ref.abs().max()))
    assert not M.conv2d_f16_supported(16, 64, 8, 8, 3, 1) and not M.conv2d_f16_supported(16, 32, 32, 32, 3, 1)
    assert not M.conv2d_f16_supported(16, 64, 16, 16, 3, 2) and M.conv2d_f16_supported(16, 64, 16, 16, 3, 1)


@pytest.mark.parametrize("stride,B,cin,cout,H,W", [(1, 2, 32, 64, 16, 32), (1, 1, 48, 64, 17, 20), (2, 2, 32, 64, 16, 64),
                                                  (2, 1, 16, 64, 31, 33), (2, 2, 32, 64, 32, 32)])
def test_conv2d_f16_presplit_input(simlib, stride, B, cin, cout, H, W):
    """PRE mode of csrc/convh_enc.hip: the input arrives split + K-blocked (hf_split_activation_f16, the
    pre-conv affine applied there) and is staged by LDS-DMA, incl. the parity-split stride-2 tile and the
    zero padding - bit-identical to the in-kernel conversion."""
    torch.manual_seed(stride + W)
    x = torch.randn(B, cin, H, W)
    w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    a, t = torch.rand(cin) + 0.5, torch.randn(cin) * 0.2
    bsh = torch.randn(cout)
    wt = M.conv_prepare(simlib, None, w)
    hi, lo = M.conv_split_weights_f16(simlib, None, wt)
    for nterms in (3, 1):
        ref = M.conv2d_f16(simlib, None, x, hi, lo, nterms, cout, stride, in_scale=a, in_shift=t, bias=bsh, act=M.ACT_LRELU, alpha=0.01)
        xs = M.split_activation_f16(simlib, None, x, a, t, want_lo=nterms == 3)
        v = (x * a.view(1, -1, 1, 1) + t.view(1, -1, 1, 1)).reshape(B, cin // 8, 8, H, W).permute(0, 1, 3, 4, 2)
        back = xs.hi.float() + (xs.lo.float() if nterms == 3 else 0)
        assert maxdiff(back, v) < (2e-6 if nterms == 3 else 2e-3) * float(v.abs().max())  # (the kernel uses one fma)
        y = M.conv2d_f16(simlib, None, xs, hi, lo, nterms, cout, stride, bias=bsh, act=M.ACT_LRELU, alpha=0.01)
        assert torch.equal(y, ref)


def test_conv2d_f16_512_pixel_tile_form(simlib):
    """The 64 x 512 tile form of csrc/convh_enc.hip (each wave all 64 channels x 64 pixels: the batched swap's
    layers), forced on a small shape through hf_debug_set_tuning: same K order, so bit-identical to the
    256-pixel form - on fp32 and pre-split inputs, with BN affines, PReLU and the residual."""
    torch.manual_seed(31)
    B, cin, cout, H, W = 2, 32, 128, 32, 64
    x = torch.randn(B, cin, H, W)
    w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    a, t = torch.rand(cin) + 0.5, torch.randn(cin) * 0.2
    g, bsh, slope = torch.rand(cout) + 0.5, torch.randn(cout) * 0.2, torch.rand(cout) * 0.5
    wt = M.conv_prepare(simlib, None, w)
    hi, lo = M.conv_split_weights_f16(simlib, None, wt)
    res = torch.randn(B, cout, H, W)
    kw = dict(out_scale=g, bias=bsh, act=M.ACT_PRELU, slope=slope, residual=res)
    xs = M.split_activation_f16(simlib, None, x, a, t)
    try:
        simlib.hf_debug_set_tuning(4)  # never the 512-pixel form
        ref = M.conv2d_f16(simlib, None, x, hi, lo, 3, cout, 1, in_scale=a, in_shift=t, **kw)
        assert simlib.hf_debug_last_path() in (601, 603)
        simlib.hf_debug_set_tuning(1 << 8)  # from 8 blocks of 512 pixels (here: 2 * 4 * 2 = 16)
        y = M.conv2d_f16(simlib, None, x, hi, lo, 3, cout, 1, in_scale=a, in_shift=t, **kw)
        assert simlib.hf_debug_last_path() == 604
        ys = M.conv2d_f16(simlib, None, xs, hi, lo, 3, cout, 1, **kw)
        assert simlib.hf_debug_last_path() == 604
    finally:
        simlib.hf_debug_set_tuning(0)
    assert torch.equal(y, ref) and torch.equal(ys, ref)
    want = F.prelu(F.conv2d(x * a.view(1, -1, 1, 1) + t.view(1, -1, 1, 1), w, padding=1) * g.view(1, -1, 1, 1) + bsh.view(1, -1, 1, 1), slope) + res
    assert maxdiff(y, want) < TOL * max(1.0, float(want.abs().max()))


def test_conv2d_f16_split_k_and_grouped_presplit(simlib):
    """Few output tiles + many input channels: split-K over the K stages with the deterministic second pass
    (epilogue incl. PReLU + residual there); grouped launch on pre-split shared / per-group inputs."""
    torch.manual_seed(21)
    B, cin, cout, H, W = 1, 256, 64, 16, 16
    assert simlib.hf_conv2d_f16_workspace_floats(B, cin, cout, H, W, 1, 1) > 0  # 2 tiles, 16 stages -> split
    x = torch.randn(B, cin, H, W)
    w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    a, t = torch.rand(cin) + 0.5, torch.randn(cin) * 0.2
    g, bsh, slope = torch.rand(cout) + 0.5, torch.randn(cout) * 0.2, torch.rand(cout) * 0.5
    wt = M.conv_prepare(simlib, None, w)
    hi, lo = M.conv_split_weights_f16(simlib, None, wt)
    ref = F.prelu(F.conv2d(x * a.view(1, -1, 1, 1) + t.view(1, -1, 1, 1), w, padding=1) * g.view(1, -1, 1, 1) + bsh.view(1, -1, 1, 1), slope)
    res = torch.randn_like(ref)
    y = M.conv2d_f16(simlib, None, x, hi, lo, 3, cout, 1, in_scale=a, in_shift=t, out_scale=g, bias=bsh, act=M.ACT_PRELU, slope=slope,
                     residual=res)
    assert maxdiff(y, ref + res) < TOL * max(1.0, float(ref.abs().max()))
    G, B, cin, cout, H, W = 2, 2, 16, 64, 32, 32
    x = torch.randn(B, cin, H, W)
    ws = [torch.randn(cout, cin, 3, 3) / 12 for _ in range(G)]
    bias = torch.randn(G, cout)
    hi, lo = M.conv_split_weights_f16(simlib, None, torch.stack([M.conv_prepare(simlib, None, w_) for w_ in ws]).contiguous())
    y = M.conv2d_f16(simlib, None, M.split_activation_f16(simlib, None, x), hi, lo, 3, cout, 2, bias=bias, act=M.ACT_LRELU, alpha=0.01,
                     groups=G, x_shared=True)
    for gi in range(G):
        assert maxdiff(y[gi], F.leaky_relu(F.conv2d(x, ws[gi], bias[gi], stride=2, padding=1), 0.01)) < TOL * 10
    x2 = torch.randn(G, B, cin, H, W)
    y2 = M.conv2d_f16(simlib, None, M.split_activation_f16(simlib, None, x2), hi, lo, 3, cout, 1, bias=bias, groups=G, x_shared=False)
    for gi in range(G):
        assert maxdiff(y2[gi], F.conv2d(x2[gi], ws[gi], bias[gi], padding=1)) < TOL * 10


def test_postprocess_modulation_module(sim_backend, golden):
    """SURVEY section 8 row f1, latent branch: ModulationModule (Linear, LayerNorm([18,512]) without affine,
    the two Linear-LayerNorm-LeakyReLU-Linear branches, x*(1+gamma)+beta, LeakyReLU unless last) on the HIP
    kernels against the reference's golden outputs; PixelNorm over the layer dim; the final axpby."""
    from hairfastgan_amd.encoders.post_process import ModulationModule
    from oracle import ref_postprocess as PP

    G = golden("postprocess.npz")
    shapes = PP.post_process_param_shapes()
    shapes.pop("latent_avg")
    P = C.params_from_shapes("pp", shapes)
    xm, em = C.unit_input("pp/mod/x", (2, 18, 512)), C.unit_input("pp/mod/e", (2, 18, 512))
    for idx, key in ((0, "pp_mod_mid"), (4, "pp_mod_last")):
        m = ModulationModule(18, idx == 4)
        m.load_state_dict({k[len(f"to_latent_1.{idx}."):]: v for k, v in P.items() if k.startswith(f"to_latent_1.{idx}.")})
        y = m(xm[:1], em[:1])  # one of the two golden samples (every op is per sample; CPU suite budget)
        ref = torch.from_numpy(G[key])[:1]
        assert maxdiff(y, ref) < 1e-4 * max(1.0, float(ref.abs().max()))
        assert maxdiff(torch.from_numpy(G[key]), PP.modulation_module(P, f"to_latent_1.{idx}", xm, em, 18, idx == 4)) == 0
    # the stack form (gamma / beta branches of all modules up front: one stacked Linear + one grouped LayerNorm): the same
    # bits as module by module, also after a load_state_dict drops the stacked copies
    from hairfastgan_amd.encoders.post_process import modulation_stack
    from torch import nn

    stack = nn.ModuleList([ModulationModule(18, i == 2) for i in range(3)])
    for i, m in enumerate(stack):
        m.load_state_dict({k[len(f"to_latent_1.{i}."):]: v for k, v in P.items() if k.startswith(f"to_latent_1.{i}.")})
    seq = xm[:1]
    for m in stack:
        seq = m(seq, em[:1])
    assert torch.equal(modulation_stack(stack, xm[:1], em[:1]), seq)
    stack[1].load_state_dict({k[len("to_latent_2.1."):]: v for k, v in P.items() if k.startswith("to_latent_2.1.")})  # not the head module
    seq = xm[:1]
    for m in stack:
        seq = m(seq, em[:1])
    assert torch.equal(modulation_stack(stack, xm[:1], em[:1]), seq)
    x = torch.randn(3, 18, 64)
    simlib = sim_backend[0].lib()
    gg, gb = torch.rand(3, 32) + 0.5, torch.randn(3, 32)
    rows6 = torch.randn(6, 32)
    want = torch.stack([F.leaky_relu(F.layer_norm(rows6[r], [32], gg[r % 3], gb[r % 3])) for r in range(6)])
    assert maxdiff(M.layernorm(simlib, None, rows6, 32, gg, gb, lrelu=True, groups=3), want) < 1e-5
    assert maxdiff(M.pixel_norm_dim1(simlib, None, x), PP.pixel_norm(x)) < 1e-6
    a, b = torch.randn(2, 18, 32), torch.randn(18, 32)
    assert maxdiff(M.axpby(simlib, None, a, 0.1, b.reshape(-1), 1.0), 0.1 * a + b) < 1e-6
    assert maxdiff(M.layernorm(simlib, None, a, 18 * 32), F.layer_norm(a, [18, 32])) < 1e-5
    g_, b_ = torch.rand(32) + 0.5, torch.randn(32)
    assert maxdiff(M.layernorm(simlib, None, a, 32, g_, b_, lrelu=True), F.leaky_relu(F.layer_norm(a, [32], g_, b_))) < 1e-5
    xl, wl4, bl4 = torch.randn(3, 4112), torch.randn(6, 4112) / 64, torch.randn(6)  # long rows: the K-split form (4 waves per row group)
    assert maxdiff(M.linear(simlib, None, xl, wl4, bl4, 1.0), F.linear(xl, wl4, bl4)) < 2e-5
    rows = torch.randn(19, 40)  # > 8 rows: several row chunks in one launch
    wl, bl = torch.randn(7, 40), torch.randn(7)
    assert maxdiff(M.linear(simlib, None, rows, wl, bl, 1.0), F.linear(rows, wl, bl)) < 1e-5


def test_postprocess_state_dict_matches_reference_layout():
    from hairfastgan_amd.encoders import PostProcessModel
    from oracle import ref_postprocess as PP

    want = PP.post_process_param_shapes()
    want.pop("latent_avg")
    got = {k: tuple(v.shape) for k, v in PostProcessModel().state_dict().items()}
    assert got == want and list(got) == list(want)


def test_latent_models_oracle_and_layout(golden):
    """SURVEY section 8 row f4, the two ModulationModule stacks: the oracle restatements of RotateModel and
    ClipBlendingModel (around a stand-in image tower) reproduce the reference's golden outputs bit for bit, and the
    HIP-backed mirrors have the reference's state-dict layout (their kernels: test_postprocess_modulation_module;
    the complete models against the goldens: tests/test_gpu_encoders.py)."""
    from hairfastgan_amd.encoders import ClipBlendingModel, RotateModel
    from oracle import ref_postprocess as PP

    G = golden("latent_models.npz")
    w_from, w_to, s_face, s_color, img_face, img_color = C.latent_model_inputs()
    Pr = C.params_from_shapes("rotate", PP.rotate_param_shapes())
    assert torch.equal(PP.rotate_model(Pr, w_from, w_to), torch.from_numpy(G["rotate"]))
    Pb = C.params_from_shapes("clipblend", PP.clip_blending_param_shapes())
    assert torch.equal(PP.clip_blending(Pb, s_face, s_color, img_face, img_color, C.fake_clip_embed), torch.from_numpy(G["clip_blend"]))
    for cls, shapes in ((RotateModel, PP.rotate_param_shapes()), (ClipBlendingModel, PP.clip_blending_param_shapes())):
        sd = cls().state_dict()
        assert {k: tuple(v.shape) for k, v in sd.items()} == shapes and list(sd) == list(shapes)
    with pytest.raises(NotImplementedError, match="CLIP"):
        ClipBlendingModel().get_image_embed(img_face)


def test_fused_conv_small_plane_routes(sim_backend, simlib):
    """encoders/_fused.conv on planes the tiled fp16 kernel does not take: stride-2 3x3 with outputs up to 8x8 as a GEMM over
    unfolded patches (grouped like the e4e style heads: shared input first, per-group inputs after), dense stride-1 3x3 as
    the tap GEMM; both on csrc/gemm_h.hip (hf_debug_last_path family 7) and equal to torch's convolution."""
    from hairfastgan_amd.encoders._fused import PreparedConv, conv

    torch.manual_seed(5)
    G, B, cin, cout = 3, 2, 256, 256
    w = torch.randn(G, cout, cin, 3, 3) / (cin * 9) ** 0.5
    bias = torch.randn(G, cout)
    wt = torch.stack([M.conv_prepare(simlib, None, w[g]) for g in range(G)]).contiguous()
    pc = PreparedConv(wt, 3)
    x = torch.randn(B, cin, 6, 6)
    y = conv(x, pc, 3, 2, bias=bias, act=M.ACT_LRELU, alpha=0.01, groups=G, x_shared=True)            # [G,B,cout,3,3]
    assert simlib.hf_debug_last_path() // 100 == 7 and tuple(y.shape) == (G, B, cout, 3, 3)
    for g in range(G):
        ref = F.leaky_relu(F.conv2d(x, w[g], bias[g], stride=2, padding=1), 0.01)
        assert maxdiff(y[g], ref) < TOL * max(1.0, float(ref.abs().max()))
    y2 = conv(y, pc, 3, 2, bias=bias, act=M.ACT_LRELU, alpha=0.01, groups=G, x_shared=False)          # 3x3 -> 2x2, own inputs
    y3 = conv(y2, pc, 3, 2, bias=bias, act=M.ACT_LRELU, alpha=0.01, groups=G, x_shared=False)         # 2x2 -> 1x1
    for g in range(G):
        r2 = F.leaky_relu(F.conv2d(y[g], w[g], bias[g], stride=2, padding=1), 0.01)
        assert maxdiff(y2[g], r2) < TOL * max(1.0, float(r2.abs().max()))
        r3 = F.leaky_relu(F.conv2d(y2[g], w[g], bias[g], stride=2, padding=1), 0.01)
        assert tuple(y3[g].shape) == (B, cout, 1, 1) and maxdiff(y3[g], r3) < TOL * max(1.0, float(r3.abs().max()))
    # dense stride-1 conv on a 4x4 plane with 512 x 512 weights: the tap GEMM
    w1 = torch.randn(512, 512, 3, 3) / (512 * 9) ** 0.5
    b1 = torch.randn(512)
    p1 = PreparedConv(M.conv_prepare(simlib, None, w1), 3)
    x1 = torch.randn(2, 512, 4, 4)
    z = conv(x1, p1, 3, 1, bias=b1)
    assert simlib.hf_debug_last_path() // 100 == 7
    ref = F.conv2d(x1, w1, b1, padding=1)
    assert maxdiff(z, ref) < TOL * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("stride,nterms", [(1, 3), (2, 3), (1, 1)])
def test_conv2d_f16_split_output(simlib, stride, nterms):
    """hf_conv2d_f16_split_f32: the conv's epilogue writes next_scale*y + next_shift in the pre-split K-blocked layout - bit
    for bit what hf_split_activation_f16 makes of the fp32 result, which is also still written when asked for."""
    torch.manual_seed(7 + stride)
    B, cin, cout, H, W = 2, 32, 64, 16, 32
    x = torch.randn(B, cin, H, W)
    w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    hi, lo = M.conv_split_weights_f16(simlib, None, M.conv_prepare(simlib, None, w))
    a, t = torch.rand(cin) + 0.5, torch.randn(cin) * 0.2
    g, bsh, slope = torch.rand(cout) + 0.5, torch.randn(cout) * 0.2, torch.rand(cout) * 0.5
    na, nt = torch.rand(cout) + 0.5, torch.randn(cout) * 0.3
    kw = dict(in_scale=a, in_shift=t, out_scale=g, bias=bsh, act=M.ACT_PRELU, slope=slope)
    assert M.conv2d_f16_split_supported(simlib, B, cin, cout, H, W, stride)
    ref = M.conv2d_f16(simlib, None, x, hi, lo, nterms, cout, stride, **kw)
    sp, y = M.conv2d_f16_split(simlib, None, x, hi, lo, nterms, cout, stride, next_scale=na, next_shift=nt, want_f32=True, **kw)
    assert torch.equal(y, ref)
    want = M.split_activation_f16(simlib, None, ref, na, nt, want_lo=nterms == 3)
    assert torch.equal(sp.hi, want.hi) and (nterms == 1 or torch.equal(sp.lo, want.lo))
    sp2, none = M.conv2d_f16_split(simlib, None, x, hi, lo, nterms, cout, stride, **kw)   # split only, identity hand-off
    want2 = M.split_activation_f16(simlib, None, ref, want_lo=nterms == 3)
    assert none is None and torch.equal(sp2.hi, want2.hi) and (nterms == 1 or torch.equal(sp2.lo, want2.lo))
    # and it feeds the next conv exactly like the two-call form
    w2 = torch.randn(64, cout, 3, 3) / (cout * 9) ** 0.5
    hi2, lo2 = M.conv_split_weights_f16(simlib, None, M.conv_prepare(simlib, None, w2))
    z1 = M.conv2d_f16(simlib, None, sp2, hi2, lo2, nterms, 64, 1)
    z2 = M.conv2d_f16(simlib, None, want2, hi2, lo2, nterms, 64, 1)
    assert torch.equal(z1, z2)


@pytest.mark.parametrize("stride,pre,nterms", [(1, False, 3), (1, True, 3), (2, True, 3), (1, True, 1)])
def test_conv2d_f16_virtual_split_k_equals_the_two_launch_form(simlib, nterms, stride, pre):
    """Batch-invariant plans (the default): the K partition of a layer is the canonical launch's (batch 3), whatever the real
    batch; a launch that fills the chip by itself walks the slabs INSIDE its blocks (ConvParams::vsplit: a second
    accumulator set, slabs added in z order, the block's own store_tile_rows epilogue) - bit for bit what the batch-1
    launch of the same layer gives with its slabs in memory and the splitk_reduce pass.  Also: the 512-pixel tile form,
    the split output (available again, since the epilogue runs in the conv kernel), PReLU + residual in the epilogue."""
    torch.manual_seed(41 + stride)
    cin, cout, H, W = 192, 64, 16 * stride, 32 * stride  # 12 K stages over 12 blocks: the canonical plan makes three slabs of them
    form512 = stride == 1 and nterms == 3 and pre  # also the 64 x 512 tile form: 13 images = 13 such tiles (> the canonical launch's 12 blocks)
    B = 13 if form512 else 6
    x = torch.randn(B, cin, H, W)
    w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    hi, lo = M.conv_split_weights_f16(simlib, None, M.conv_prepare(simlib, None, w))
    a, t = torch.rand(cin) + 0.5, torch.randn(cin) * 0.2
    g, bsh, slope = torch.rand(cout) + 0.5, torch.randn(cout) * 0.2, torch.rand(cout) * 0.5
    res = torch.randn(B, cout, H // stride, W // stride)
    kw = dict(out_scale=g, bias=bsh, act=M.ACT_PRELU, slope=slope)
    src = lambda xx: (M.split_activation_f16(simlib, None, xx, a, t, want_lo=nterms == 3), {}) if pre else (xx, dict(in_scale=a, in_shift=t))  # noqa: E731
    prev = simlib.hf_set_batch_invariant(1)
    try:
        # one sample at a time: 4 (stride 2: 4) output tiles x 16 stages -> the canonical plan splits K, for real
        assert simlib.hf_conv2d_f16_workspace_floats(1, cin, cout, H, W, stride, 1) > 0
        assert not M.conv2d_f16_split_supported(simlib, 1, cin, cout, H, W, stride, nterms, pre=pre)
        singles = []
        for b in range(B):
            xin, kin = src(x[b:b + 1])
            singles.append(M.conv2d_f16(simlib, None, xin, hi, lo, nterms, cout, stride, residual=res[b:b + 1], **kw, **kin))
        ref = torch.cat(singles)
        # the whole batch (>= 24 output tiles) with "the chip" shrunk to 13 blocks - the canonical batch-3 launch (12 tiles) still
        # splits, this one fills it: virtual split, no workspace, same bits
        simlib.hf_debug_set_tuning(13 << 24)
        assert simlib.hf_conv2d_f16_workspace_floats(B, cin, cout, H, W, stride, 1) == 0  # (the query plans register-staged input)
        xin, kin = src(x)
        y = M.conv2d_f16(simlib, None, xin, hi, lo, nterms, cout, stride, residual=res, **kw, **kin)
        assert torch.equal(y, ref)
        if form512:  # (pre-split input only: the register-staged 512-pixel form has no registers left for the second accumulator set)
            simlib.hf_debug_set_tuning((13 << 24) | (1 << 8))
            y512 = M.conv2d_f16(simlib, None, xin, hi, lo, nterms, cout, stride, residual=res, **kw, **kin)
            assert simlib.hf_debug_last_path() == 604 and torch.equal(y512, ref)
            simlib.hf_debug_set_tuning(13 << 24)
        # the split output of the virtual form = the split of the fp32 result
        assert M.conv2d_f16_split_supported(simlib, B, cin, cout, H, W, stride, nterms, pre=pre)
        sp, y2 = M.conv2d_f16_split(simlib, None, xin, hi, lo, nterms, cout, stride, want_f32=True, **kw, **kin)
        noresid = torch.cat([M.conv2d_f16(simlib, None, src(x[b:b + 1])[0], hi, lo, nterms, cout, stride, **kw, **src(x[b:b + 1])[1])
                             for b in range(B)])
        assert torch.equal(y2, noresid)
        want = M.split_activation_f16(simlib, None, noresid, want_lo=nterms == 3)
        assert torch.equal(sp.hi, want.hi) and (nterms == 1 or torch.equal(sp.lo, want.lo))
    finally:
        simlib.hf_debug_set_tuning(0)
        simlib.hf_set_batch_invariant(prev)
    ref32 = F.prelu(F.conv2d(x * a.view(1, -1, 1, 1) + t.view(1, -1, 1, 1), w, stride=stride, padding=1) * g.view(1, -1, 1, 1)
                    + bsh.view(1, -1, 1, 1), slope) + res
    assert maxdiff(y, ref32) < (TOL if nterms == 3 else 4e-3) * max(1.0, float(ref32.abs().max()))


def test_conv1x1_f16_virtual_split_k_equals_the_two_launch_form(simlib):
    """gemm_h.hip under batch-invariant plans: hf_conv1x1_f16_f32 of a batch that fills "the chip" runs the canonical K
    partition inside its blocks - the bits of the one-sample launches with their slabs and splitk_reduce; and the tap-GEMM
    of the small-plane modulated conv (raw slabs + small_combine) likewise."""
    torch.manual_seed(43)
    B, cin, cout, H, W = 8, 512, 64, 8, 8
    x = torch.randn(B, cin, H, W)
    w = torch.randn(cout, cin, 1, 1) / cin ** 0.5
    a, t = torch.rand(cin) + 0.5, torch.randn(cin) * 0.2
    g, bsh = torch.rand(cout) + 0.5, torch.randn(cout) * 0.2
    hi, lo = M.conv_split_weights_f16(simlib, None, M.conv_prepare(simlib, None, w))
    prev = simlib.hf_set_batch_invariant(1)
    try:
        assert simlib.hf_conv1x1_f16_workspace_floats(1, cin, cout, H, W, 1, 1) > 0
        kw = dict(in_scale=a, in_shift=t, out_scale=g, bias=bsh, act=M.ACT_LRELU, alpha=0.2)
        ref = torch.cat([M.conv1x1_f16(simlib, None, x[b:b + 1], hi, lo, 3, cout, 1, **kw) for b in range(B)])
        simlib.hf_debug_set_tuning(3 << 24)  # canonical batch 3: 2 blocks (two 8 x 8 images per tile) - splits; batch 8: 4 blocks
        assert simlib.hf_conv1x1_f16_workspace_floats(B, cin, cout, H, W, 1, 1) == 0
        y = M.conv1x1_f16(simlib, None, x, hi, lo, 3, cout, 1, **kw)
        assert torch.equal(y, ref)
    finally:
        simlib.hf_debug_set_tuning(0)
        simlib.hf_set_batch_invariant(prev)
    want = F.leaky_relu(F.conv2d(x * a.view(1, -1, 1, 1) + t.view(1, -1, 1, 1), w) * g.view(1, -1, 1, 1) + bsh.view(1, -1, 1, 1), 0.2)
    assert maxdiff(y, want) < TOL * max(1.0, float(want.abs().max()))


def test_unit_chain_hand_over_equals_unit_after_unit(sim_backend, simlib, monkeypatch):
    """Round 5, unit -> unit hand-off: the tail of a residual unit (IR-SE: hf_scale_shortcut_add_split_f16; IBasicBlock: the second
    conv's epilogue with bn3 + residual) also writes the NEXT unit's first-conv input pre-split with that unit's BatchNorm -
    no split pass at the boundary, the next conv stages by LDS-DMA.  Bit for bit the result of running unit after unit, across a
    stride-2 unit with a conv shortcut, and the hand-over really happens."""
    import sys

    from torch import nn

    from hairfastgan_amd.encoders.e4e import bottleneck_IR_SE
    from hairfastgan_amd.encoders.fs_encoder import IBasicBlock, run_block_chain

    fused = sys.modules["hairfastgan_amd.encoders._fused"]
    torch.manual_seed(3)

    def randomize(mod):
        for name, p_ in mod.named_parameters():
            p_.data.normal_(0, 0.2)
        for name, b_ in mod.named_buffers():
            if name.endswith("running_var"):
                b_.uniform_(0.5, 1.5)
            elif name.endswith("running_mean"):
                b_.normal_(0, 0.2)
        return mod.eval()

    x = torch.randn(2, 64, 32, 32)
    # --- IR-SE units (e4e): 64 -> 64, 64 -> 128 stride 2 (conv shortcut), 128 -> 128
    units = [randomize(bottleneck_IR_SE(64, 64, 1)), randomize(bottleneck_IR_SE(64, 128, 2)), randomize(bottleneck_IR_SE(128, 128, 1))]
    calls = []
    real = M.scale_shortcut_add_split
    monkeypatch.setattr(M, "scale_shortcut_add_split", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    with torch.inference_mode():
        ref = x
        for u in units:
            ref = u(ref)
        assert not calls
        y, xs = x, None
        for i, u in enumerate(units):
            nxt = units[i + 1] if i + 1 < len(units) else None
            oh, ow = (y.shape[2] - 1) // u.stride + 1, (y.shape[3] - 1) // u.stride + 1
            y, xs = u.forward_chain(y, xs, nxt if (nxt is not None and nxt.takes_split(oh, ow)) else None)
    assert len(calls) == 2 and xs is None and torch.equal(y, ref)
    # the tail kernel alone against its two-pass form
    r, gate, sc = torch.randn(2, 64, 16, 16), torch.rand(2, 64), torch.randn(2, 64, 32, 32)
    a, t = torch.rand(64) + 0.5, torch.randn(64) * 0.3
    out, sp = real(simlib, None, r, gate, sc, 2, a, t)
    want = M.scale_shortcut_add(simlib, None, r, gate, sc, 2)
    wsp = M.split_activation_f16(simlib, None, want, a, t)
    assert torch.equal(out, want) and torch.equal(sp.hi, wsp.hi) and torch.equal(sp.lo, wsp.lo)
    # --- IBasicBlocks (FS encoder / PostProcess trunk): 64 -> 64, 64 -> 64 stride 2 with its downsample branch, 64 -> 64 (four K
    # stages: no launch here spreads its K loop over the grid - such a launch cannot write the hand-over and the chain falls back)
    ds = nn.Sequential(nn.Conv2d(64, 64, 1, 2, bias=False), nn.BatchNorm2d(64, eps=1e-05))
    blocks = [randomize(IBasicBlock(64, 64)), randomize(IBasicBlock(64, 64, 2, ds)), randomize(IBasicBlock(64, 64))]
    handed = []
    real_split = M.conv2d_f16_split
    monkeypatch.setattr(M, "conv2d_f16_split", lambda *a, **k: (handed.append(k.get("want_f32", False)), real_split(*a, **k))[1])
    with torch.inference_mode():
        monkeypatch.setattr(fused, "USE_CHAIN", False)
        ref = x
        for b_ in blocks:
            ref = b_(ref)
        assert not any(handed)
        monkeypatch.setattr(fused, "USE_CHAIN", True)
        y, xs = run_block_chain(blocks, x)
    assert sum(handed) == 2 and xs is None and torch.equal(y, ref)


def test_conv2d_f16_block_order_does_not_change_results(simlib):
    """hf_conv2d_f16_f32 launches its grid tiles-fastest or columns-fastest (ConvParams::swap_xy: which of the input tile and the
    weight column stays in L2) - forced here through hf_debug_set_tuning bit 3: identical results for a plain launch, a
    grouped launch on a shared pre-split input (the e4e style heads) and a stride-2 launch."""
    torch.manual_seed(51)
    for B, cin, cout, H, W, stride, G in ((2, 32, 128, 16, 32, 1, 1), (2, 16, 64, 32, 32, 2, 3), (1, 32, 192, 32, 32, 2, 1)):
        x = torch.randn(B, cin, H, W)
        ws = torch.randn(G, cout, cin, 3, 3) / (cin * 9) ** 0.5
        wt = torch.stack([M.conv_prepare(simlib, None, ws[g]) for g in range(G)]).contiguous()
        hi, lo = M.conv_split_weights_f16(simlib, None, wt if G > 1 else wt[0])
        bias = torch.randn(G, cout) if G > 1 else torch.randn(cout)
        xs = M.split_activation_f16(simlib, None, x)
        kw = dict(bias=bias, act=M.ACT_LRELU, alpha=0.01, groups=G)
        try:
            simlib.hf_debug_set_tuning(0)
            ref = M.conv2d_f16(simlib, None, xs, hi, lo, 3, cout, stride, **kw)
            simlib.hf_debug_set_tuning(8)
            y = M.conv2d_f16(simlib, None, xs, hi, lo, 3, cout, stride, **kw)
            y32 = M.conv2d_f16(simlib, None, x, hi, lo, 3, cout, stride, **kw)
        finally:
            simlib.hf_debug_set_tuning(0)
        assert torch.equal(y, ref) and torch.equal(y32, ref)
        for g in range(G):
            want = F.leaky_relu(F.conv2d(x, ws[g], (bias[g] if G > 1 else bias), stride=stride, padding=1), 0.01)
            assert maxdiff(y[g] if G > 1 else y, want) < TOL * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("nterms", [3, 1])
def test_conv2d_f16_stride2_multi_tile_form(simlib, nterms):
    """conv_enc_s2mt_h (round 5): a stride-2 block keeps a K stage's weights for four consecutive 128-pixel tiles (four
    accumulator tiles per wave, the next step's activations and a share of the next chunk's weights arriving meanwhile) - the
    bits of the one-tile form: ragged planes (tiles past the end, halo units outside the image that differ from tile to
    tile), two images, a grouped launch on per-group inputs, a virtual split-K, PReLU + residual, the split output."""
    torch.manual_seed(61)
    # f4 / f2: "chips" (blocks per round) that the launch's blocks / 4 and / 2 fill in whole rounds (20 and 32 one-tile blocks)
    for B, cin, cout, H, W, G, f4, f2 in ((2, 48, 64, 40, 72, 1, 5, 10), (1, 32, 128, 64, 64, 2, 4, 6)):
        x = torch.randn(*((G, B) if G > 1 else (B,)), cin, H, W)
        ws = torch.randn(G, cout, cin, 3, 3) / (cin * 9) ** 0.5
        wt = torch.stack([M.conv_prepare(simlib, None, ws[g]) for g in range(G)]).contiguous()
        hi, lo = M.conv_split_weights_f16(simlib, None, wt if G > 1 else wt[0])
        bias = torch.randn(G, cout) if G > 1 else torch.randn(cout)
        slope = torch.rand(G, cout) if G > 1 else torch.rand(cout)
        oh, ow = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        res = torch.randn(*((G, B) if G > 1 else (B,)), cout, oh, ow)
        xs = M.split_activation_f16(simlib, None, x, want_lo=nterms == 3)
        kw = dict(bias=bias, act=M.ACT_PRELU, slope=slope, residual=res, groups=G, x_shared=False)
        try:
            simlib.hf_debug_set_tuning(0)
            ref = M.conv2d_f16(simlib, None, xs, hi, lo, nterms, cout, 2, **kw)
            assert simlib.hf_debug_last_path() == 602
            simlib.hf_debug_set_tuning(f2 << 24)
            yb = M.conv2d_f16(simlib, None, xs, hi, lo, nterms, cout, 2, **kw)
            assert simlib.hf_debug_last_path() == 606 and torch.equal(yb, ref)
            simlib.hf_debug_set_tuning(f4 << 24)
            y = M.conv2d_f16(simlib, None, xs, hi, lo, nterms, cout, 2, **kw)
            assert simlib.hf_debug_last_path() == 605
            if G == 1:
                sp, y2 = M.conv2d_f16_split(simlib, None, xs, hi, lo, nterms, cout, 2, want_f32=True, bias=bias, act=M.ACT_PRELU, slope=slope)
                assert simlib.hf_debug_last_path() == 605
        finally:
            simlib.hf_debug_set_tuning(0)
        assert torch.equal(y, ref)
        if G == 1:
            noresid = M.conv2d_f16(simlib, None, xs, hi, lo, nterms, cout, 2, bias=bias, act=M.ACT_PRELU, slope=slope)
            want = M.split_activation_f16(simlib, None, noresid, want_lo=nterms == 3)
            assert torch.equal(y2, noresid) and torch.equal(sp.hi, want.hi) and (nterms == 1 or torch.equal(sp.lo, want.lo))
    # virtual split-K inside the multi-tile form (batch-invariant plans): 12 K stages, canonical batch 3 splits them
    B, cin, cout, H, W = 13, 192, 64, 32, 64
    x = torch.randn(B, cin, H, W)
    w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    hi, lo = M.conv_split_weights_f16(simlib, None, M.conv_prepare(simlib, None, w))
    bias = torch.randn(cout)
    prev = simlib.hf_set_batch_invariant(1)
    try:
        assert simlib.hf_conv2d_f16_workspace_floats(1, cin, cout, H, W, 2, 1) > 0
        ref = torch.cat([M.conv2d_f16(simlib, None, M.split_activation_f16(simlib, None, x[b:b + 1], want_lo=nterms == 3), hi, lo, nterms,
                                      cout, 2, bias=bias) for b in range(B)])
        simlib.hf_debug_set_tuning(13 << 24)  # canonical launch: 3 x 4 tiles = 12 blocks, split; this one: 52 blocks / 4 = 13, one full round: virtual, x4
        xs = M.split_activation_f16(simlib, None, x, want_lo=nterms == 3)
        y = M.conv2d_f16(simlib, None, xs, hi, lo, nterms, cout, 2, bias=bias)
        assert simlib.hf_debug_last_path() == 605
        # a "chip" of 26 blocks per round: 13 four-tile blocks do not fill it, 26 two-tile blocks are exactly one round -> x2
        simlib.hf_debug_set_tuning(26 << 24)
        y2t = M.conv2d_f16(simlib, None, xs, hi, lo, nterms, cout, 2, bias=bias)
        assert simlib.hf_debug_last_path() == 606
        assert torch.equal(y2t, y)
    finally:
        simlib.hf_debug_set_tuning(0)
        simlib.hf_set_batch_invariant(prev)
    assert torch.equal(y, ref)


def test_conv2d_f16_persistent_blocks_walk_the_grid(simlib):
    """conv_enc_h's persistent form (round 5): `hf_debug_set_persistent_blocks(3)`: three resident blocks walk the plain form's grid in
    its dispatch order, tile after tile through the same stage buffers - the bits of the one-block-per-tile launch: the 512-pixel
    form on fp32 and on pre-split input (ragged planes: border tiles follow interior ones, so halo units that were data become
    zero padding and back), a grouped launch in the columns-fastest order, the stride-2 one-tile form, the split output."""
    torch.manual_seed(77)
    try:
        for B, cin, cout, H, W, stride, G, tuning in ((2, 32, 128, 24, 72, 1, 1, 1 << 8), (2, 16, 64, 32, 32, 1, 3, (1 << 8) | 8),
                                                     (2, 32, 64, 40, 40, 2, 1, 0), (5, 16, 64, 16, 32, 1, 1, 0)):
            x = torch.randn(*((G, B) if G > 1 else (B,)), cin, H, W)
            ws = torch.randn(G, cout, cin, 3, 3) / (cin * 9) ** 0.5
            wt = torch.stack([M.conv_prepare(simlib, None, ws[g]) for g in range(G)]).contiguous()
            hi, lo = M.conv_split_weights_f16(simlib, None, wt if G > 1 else wt[0])
            bias = torch.randn(G, cout) if G > 1 else torch.randn(cout)
            slope = torch.rand(G, cout) if G > 1 else torch.rand(cout)
            xs = M.split_activation_f16(simlib, None, x)
            kw = dict(bias=bias, act=M.ACT_PRELU, slope=slope, groups=G, x_shared=False)
            simlib.hf_debug_set_tuning(tuning)
            outs = {}
            for blocks in (0, 3):  # 0: the default 256 resident blocks - more than these launches have: one block per tile
                simlib.hf_debug_set_persistent_blocks(blocks)
                y32 = M.conv2d_f16(simlib, None, x, hi, lo, 3, cout, stride, **kw) if stride == 1 else None
                ys = M.conv2d_f16(simlib, None, xs, hi, lo, 3, cout, stride, **kw)
                path = simlib.hf_debug_last_path()
                sp = None
                if G == 1:
                    sp, _ = M.conv2d_f16_split(simlib, None, xs, hi, lo, 3, cout, stride, want_f32=True, bias=bias, act=M.ACT_PRELU, slope=slope)
                outs[blocks] = (y32, ys, sp, path)
            ref = outs[0]
            assert ref[3] == (604 if tuning & (1 << 8) else 602 if stride == 2 else ref[3])
            for blocks in (3,):
                y32, ys, sp, path = outs[blocks]
                assert path == ref[3]
                assert torch.equal(ys, ref[1]) and (y32 is None or torch.equal(y32, ref[0]))
                if sp is not None:
                    assert torch.equal(sp.hi, ref[2].hi) and torch.equal(sp.lo, ref[2].lo)
            for g in range(G):
                want = F.prelu(F.conv2d(x[g] if G > 1 else x, ws[g], (bias[g] if G > 1 else bias), stride=stride, padding=1), slope[g] if G > 1 else slope)
                got = ref[1][g] if G > 1 else ref[1]
                assert maxdiff(got, want) < TOL * max(1.0, float(want.abs().max()))
    finally:
        simlib.hf_debug_set_tuning(0)
        simlib.hf_debug_set_persistent_blocks(0)
