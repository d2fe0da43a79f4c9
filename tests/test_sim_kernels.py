"""CPU tests: the HIP kernel SOURCES interpreted by tests/hipsim vs the oracle and the
golden vectors produced by the real reference.  They validate index arithmetic, LDS
layouts, the MFMA fragment maps and tile geometry before any GPU time is spent; the
numerical parity proper is re-established on the GPU by the -m gpu tests."""
import math

import numpy as np
import pytest
import torch

from hairfastgan_amd import _marshal as M
from oracle import cases as C
from oracle import ref_stylegan2 as O

TOL = 2e-5


def maxdiff(a, b):
    return float((a.double() - b.double()).abs().max())


@pytest.mark.parametrize("name", list(C.UPFIRDN_CASES))
def test_upfirdn2d(simlib, golden, name):
    c = C.UPFIRDN_CASES[name]
    x, k = C.upfirdn_input(name), C.blur_kernel4()
    y = M.upfirdn2d(simlib, None, x, k, c["up"], c["up"], c["down"], c["down"], c["pad"][0], c["pad"][1],
                    c["pad"][0], c["pad"][1])
    ref = torch.from_numpy(golden("upfirdn2d.npz")[name])
    assert y.shape == ref.shape
    assert maxdiff(y, ref) < TOL


@pytest.mark.parametrize("name", list(C.ACT_CASES))
def test_fused_bias_act(simlib, golden, name):
    x, b = C.act_inputs(name)
    y = M.fused_bias_act(simlib, None, x, b, 0.2, 2 ** 0.5)
    assert maxdiff(y, torch.from_numpy(golden("fused_act.npz")[name])) < 1e-6


def test_noise_bias_act(simlib):
    x = torch.randn(2, 5, 6, 6)
    nz = torch.randn(1, 1, 6, 6)
    nw = torch.tensor([0.3])
    b = torch.randn(5)
    y = M.noise_bias_act(simlib, None, x, nz, nw, b)
    assert maxdiff(y, O.fused_leaky_relu(x + nw * nz, b)) < 1e-6
    nzb = torch.randn(2, 1, 6, 6)
    y = M.noise_bias_act(simlib, None, x, nzb, nw, b)
    assert maxdiff(y, O.fused_leaky_relu(x + nw * nzb, b)) < 1e-6
    x = torch.randn(2, 3, 5, 3)  # hw not a multiple of 4 -> scalar kernel
    nz = torch.randn(2, 1, 5, 3)
    y = M.noise_bias_act(simlib, None, x, nz, nw, b[:3].contiguous())
    assert maxdiff(y, O.fused_leaky_relu(x + nw * nz, b[:3])) < 1e-6


def _style(simlib, P, w):
    wt, wsq = M.prepare_weights(simlib, None, P["L.conv.weight"])
    s = M.modulation(simlib, None, w, P["L.conv.modulation.weight"], P["L.conv.modulation.bias"])
    return wt, wsq, s


@pytest.mark.parametrize("name", [c[0] for c in C.MODCONV_SMALL])
def test_style_and_demod(simlib, name):
    d = C.modconv_small_inputs(name)
    P = d["P_up0"]
    wt, wsq, s = _style(simlib, P, d["w"])
    s_ref = O.equal_linear(d["w"], P["L.conv.modulation.weight"], P["L.conv.modulation.bias"])
    assert maxdiff(s, s_ref) < 1e-5
    cin = d["cin"]
    wref = P["L.conv.weight"][0] / (cin * 9) ** 0.5
    assert maxdiff(wt, wref.permute(2, 3, 1, 0).reshape(9, cin, -1)) < 1e-7
    dm = M.demod(simlib, None, s, wsq)
    dref = torch.rsqrt(((wref[None] * s_ref[:, None, :, None, None]) ** 2).sum([2, 3, 4]) + 1e-8)
    assert maxdiff(dm, dref) / float(dref.abs().max()) < 1e-5


@pytest.mark.parametrize("name", [c[0] for c in C.MODCONV_SMALL])
@pytest.mark.parametrize("up", [False, True])
def test_styled_conv(simlib, golden, name, up):
    d = C.modconv_small_inputs(name)
    P = d[f"P_up{int(up)}"]
    wt, wsq, s = _style(simlib, P, d["w"])
    dm = M.demod(simlib, None, s, wsq)
    G = golden("modconv_small.npz")
    if up:
        y_conv = M.modconv3x3_up(simlib, None, d["x"], wt, s, dm, P["L.conv.blur.kernel"], None, None, None)
        y = M.modconv3x3_up(simlib, None, d["x"], wt, s, dm, P["L.conv.blur.kernel"], d["noise_up1"],
                            P["L.noise.weight"], P["L.activate.bias"])
    else:
        y_conv = M.modconv3x3(simlib, None, d["x"], wt, s, dm, None, None, None)
        y = M.modconv3x3(simlib, None, d["x"], wt, s, dm, d["noise_up0"], P["L.noise.weight"], P["L.activate.bias"])
    ref_conv = torch.from_numpy(G[f"{name}_up{int(up)}_conv"])
    ref = torch.from_numpy(G[f"{name}_up{int(up)}_styled"])
    assert y_conv.shape == ref_conv.shape
    assert maxdiff(y_conv, ref_conv) < TOL * max(1.0, float(ref_conv.abs().max()))
    assert maxdiff(y, ref) < TOL * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("name", [c[0] for c in C.MODCONV_SMALL])
@pytest.mark.parametrize("use_skip", [False, True])
def test_torgb(simlib, golden, name, use_skip):
    d = C.modconv_small_inputs(name)
    P = d["P_rgb"]
    wt, _ = M.prepare_weights(simlib, None, P["L.conv.weight"])
    s = M.modulation(simlib, None, d["w"], P["L.conv.modulation.weight"], P["L.conv.modulation.bias"])
    y = M.torgb(simlib, None, d["x_rgb"], wt, s, P["L.bias"], d["skip"] if use_skip else None,
                P["L.upsample.kernel"])
    ref = torch.from_numpy(golden("modconv_small.npz")[f"{name}_rgb_skip{int(use_skip)}"])
    assert maxdiff(y, ref) < TOL * max(1.0, float(ref.abs().max()))


def test_modconv_tile_geometries(simlib):
    """Shapes that exercise every tile family: multi-image tiles (4x4), >1 tile per
    plane with ragged edges, cout not a multiple of 32 / of 4, cin not a multiple of 8."""
    torch.manual_seed(1)
    for (B, cin, cout, H, W) in [(3, 8, 8, 4, 4), (1, 12, 34, 40, 72), (2, 16, 33, 9, 5), (5, 8, 64, 2, 2), (1, 8, 8, 1, 1), (2, 8, 40, 70, 1)]:
        x = torch.randn(B, cin, H, W)
        wgt = torch.randn(1, cout, cin, 3, 3)
        mw, mb, sty = torch.randn(cin, 16), torch.randn(cin), torch.randn(B, 16)
        wt, wsq = M.prepare_weights(simlib, None, wgt)
        s = M.modulation(simlib, None, sty, mw, mb)
        dm = M.demod(simlib, None, s, wsq)
        for up in (False, True):
            ref = O.modulated_conv2d(x, sty, wgt, mw, mb, True, up)
            if up:
                y = M.modconv3x3_up(simlib, None, x, wt, s, dm, O.blur_kernel_1d_to_2d(gain=4.0), None, None, None)
            else:
                y = M.modconv3x3(simlib, None, x, wt, s, dm, None, None, None)
            assert y.shape == ref.shape
            assert maxdiff(y, ref) < TOL * max(1.0, float(ref.abs().max())), (B, cin, cout, H, W, up)


@pytest.mark.parametrize("cfg,shape", [
    (11, (2, 16, 128, 16, 32)),   # 128 co x 256 px, 8 waves
    (12, (1, 16, 64, 24, 40)),    # 64 co x 256 px, ragged plane
    (13, (2, 8, 32, 16, 16)),     # 32 co x 256 px, 16-wide rows
    (14, (1, 16, 128, 8, 32)),
    (15, (2, 8, 64, 8, 8)),
    (16, (1, 8, 32, 32, 32)),
    (31, (2, 16, 128, 16, 32)),   # DMA-staged variants (rows of 32 pixels)
    (32, (1, 16, 64, 24, 64)),
    (33, (2, 8, 32, 8, 96)),
    (34, (1, 16, 128, 12, 32)),   # ragged: 12 rows with 4-row tiles
    (21, (2, 16, 64, 8, 32)),     # up: 64 co x 128 px x 4 phases (+ rim launch)
    (22, (1, 8, 32, 16, 32)),
    (23, (1, 8, 64, 6, 40)),
    (24, (1, 8, 64, 16, 32)),
    (25, (2, 8, 32, 4, 32)),
])
def test_modconv_pipelined_configs(simlib, cfg, shape):
    """Every instantiation of the double-buffered (global_load_lds + register prefetch) kernel."""
    B, cin, cout, H, W = shape
    torch.manual_seed(cfg)
    x = torch.randn(B, cin, H, W)
    wgt = torch.randn(1, cout, cin, 3, 3)
    mw, mb, sty = torch.randn(cin, 16), torch.randn(cin), torch.randn(B, 16)
    wt, wsq = M.prepare_weights(simlib, None, wgt)
    s = M.modulation(simlib, None, sty, mw, mb)
    dm = M.demod(simlib, None, s, wsq)
    up = 20 <= cfg < 30
    ref = O.modulated_conv2d(x, sty, wgt, mw, mb, True, up)
    try:
        simlib.hf_debug_set_dispatch(0 if up else cfg, cfg if up else 0)
        if up:
            y = M.modconv3x3_up(simlib, None, x, wt, s, dm, O.blur_kernel_1d_to_2d(gain=4.0), None, None, None)
        else:
            y = M.modconv3x3(simlib, None, x, wt, s, dm, None, None, None)
        assert simlib.hf_debug_last_path() == 200 + cfg, "shape fell back from the pipelined kernel"
    finally:
        simlib.hf_debug_set_dispatch(0, 0)
    assert y.shape == ref.shape
    assert maxdiff(y, ref) < TOL * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("nterms,tol", [(3, 5e-6), (1, 4e-3)])
@pytest.mark.parametrize("cfg,shape", [(51, (2, 32, 64, 8, 32)), (51, (1, 48, 128, 20, 70)), (52, (1, 48, 64, 20, 40)), (53, (1, 16, 32, 16, 64))])
def test_modconv_f16_matrix_cores(simlib, nterms, tol, cfg, shape):
    """csrc/convh.hip: fp16 MFMA with split (hi, lo) operands reproduces the fp32 result to
    fp32-class accuracy; with plain fp16 operands to fp16 operand rounding.  Odd chunk count,
    ragged rows/columns, noise + bias + lrelu epilogue."""
    B, cin, cout, H, W = shape
    torch.manual_seed(7)
    x = torch.randn(B, cin, H, W)
    wgt = torch.randn(1, cout, cin, 3, 3)
    mw, mb, sty = torch.randn(cin, 16), torch.randn(cin), torch.randn(B, 16)
    nz, nw, bias = torch.randn(B, 1, H, W), torch.tensor([0.3]), torch.randn(cout)
    wt, wsq = M.prepare_weights(simlib, None, wgt)
    s = M.modulation(simlib, None, sty, mw, mb)
    dm = M.demod(simlib, None, s, wsq)
    hi, lo = M.split_weights_f16(simlib, None, wt)
    # hi + lo carries 2^k * weight (k: power-of-two pre-scale putting max|w| at 2^13) to 2^-22 relative
    # of the LARGEST weight - and, because of the pre-scale, of every weight above 2^-15 of it
    unscale = M.split_weights_unscale(hi)
    assert 2 ** 13 <= float(wt.abs().max()) / unscale < 2 ** 14 and math.log2(unscale) == round(math.log2(unscale))
    back = (hi.float() + lo.float()).permute(1, 0, 2, 4, 3).reshape(9, cin, cout) * unscale
    assert maxdiff(back, wt) < 3e-7 * float(wt.abs().max())
    rel = ((back.double() - wt.double()).abs() / wt.double().abs().clamp_min(1e-30))[wt.abs() > 2.0 ** -14 * wt.abs().max()]
    assert float(rel.max()) < 2.0 ** -21
    ref = M.modconv3x3(simlib, None, x, wt, s, dm, nz, nw, bias)
    try:
        simlib.hf_debug_set_dispatch(cfg, 0)
        y = M.modconv3x3_f16(simlib, None, x, hi, lo, nterms, s, dm, nz, nw, bias)
        assert simlib.hf_debug_last_path() == 500 + cfg
    finally:
        simlib.hf_debug_set_dispatch(0, 0)
    assert y.shape == ref.shape
    assert maxdiff(y, ref) < tol * max(1.0, float(ref.abs().max()))
    if nterms == 3:
        full = O.fused_leaky_relu(O.modulated_conv2d(x, sty, wgt, mw, mb, True, False) + nw * nz, bias)
        assert maxdiff(y, full) < TOL * max(1.0, float(full.abs().max()))


@pytest.mark.parametrize("nterms,tol", [(3, 5e-6), (1, 4e-3)])
@pytest.mark.parametrize("cfg,shape", [(61, (1, 32, 64, 16, 16)), (61, (2, 16, 64, 9, 40)), (63, (1, 16, 32, 16, 32))])
def test_modconv_up_f16_matrix_cores(simlib, nterms, tol, cfg, shape):
    """Transposed conv on the fp16 matrix cores: interior tiles plus the rim row / column
    families, against the fp32 MFMA kernel and the oracle (incl. blur + noise + bias + lrelu)."""
    B, cin, cout, H, W = shape
    torch.manual_seed(11)
    x = torch.randn(B, cin, H, W)
    wgt = torch.randn(1, cout, cin, 3, 3)
    mw, mb, sty = torch.randn(cin, 16), torch.randn(cin), torch.randn(B, 16)
    nz, nw, bias = torch.randn(B, 1, 2 * H, 2 * W), torch.tensor([0.3]), torch.randn(cout)
    wt, wsq = M.prepare_weights(simlib, None, wgt)
    s = M.modulation(simlib, None, sty, mw, mb)
    dm = M.demod(simlib, None, s, wsq)
    hi, lo = M.split_weights_f16(simlib, None, wt)
    k4 = O.blur_kernel_1d_to_2d(gain=4.0)
    assert M.modconv3x3_up_f16_supported(cin, cout, H, W)
    ref = M.modconv3x3_up(simlib, None, x, wt, s, dm, k4, nz, nw, bias)
    y = M.modconv3x3_up(simlib, None, x, wt, s, dm, k4, nz, nw, bias, f16=(hi, lo, nterms))
    assert simlib.hf_debug_last_path() == 500 + cfg
    assert y.shape == ref.shape
    assert maxdiff(y, ref) < tol * max(1.0, float(ref.abs().max()))
    if nterms == 3:
        full = O.fused_leaky_relu(O.modulated_conv2d(x, sty, wgt, mw, mb, True, True) + nw * nz, bias)
        assert maxdiff(y, full) < TOL * max(1.0, float(full.abs().max()))


@pytest.mark.parametrize("up", [False, True])
@pytest.mark.parametrize("blocks", [1, 3, 5])
def test_modconv_f16_persistent_tile_walk(simlib, up, blocks):
    """One resident block walks several tiles as one pipeline (csrc/convh.hip): tile-to-tile
    hand-over across rows, images (the second s slot) and - transposed conv - rim families,
    with a block count that does not divide the tile count."""
    B, cin, cout, H, W = (3, 32, 64, 16, 32) if not up else (2, 32, 64, 16, 32)
    torch.manual_seed(5)
    x = torch.randn(B, cin, H, W)
    wgt = torch.randn(1, cout, cin, 3, 3)
    mw, mb, sty = torch.randn(cin, 16), torch.randn(cin), torch.randn(B, 16)
    oh, ow = (2 * H, 2 * W) if up else (H, W)
    nz, nw, bias = torch.randn(B, 1, oh, ow), torch.tensor([0.3]), torch.randn(cout)
    wt, wsq = M.prepare_weights(simlib, None, wgt)
    s = M.modulation(simlib, None, sty, mw, mb)
    dm = M.demod(simlib, None, s, wsq)
    hi, lo = M.split_weights_f16(simlib, None, wt)
    k4 = O.blur_kernel_1d_to_2d(gain=4.0)
    try:
        simlib.hf_debug_set_persistent_blocks(blocks)
        if up:
            y = M.modconv3x3_up(simlib, None, x, wt, s, dm, k4, nz, nw, bias, f16=(hi, lo, 3))
        else:
            y = M.modconv3x3_f16(simlib, None, x, hi, lo, 3, s, dm, nz, nw, bias)
    finally:
        simlib.hf_debug_set_persistent_blocks(0)
    full = O.fused_leaky_relu(O.modulated_conv2d(x, sty, wgt, mw, mb, True, up) + nw * nz, bias)
    assert maxdiff(y, full) < TOL * max(1.0, float(full.abs().max()))


@pytest.mark.parametrize("shape", [(2, 32, 64, 16, 32), (1, 16, 32, 20, 40), (1, 32, 128, 16, 32), (1, 16, 96, 16, 32)])
def test_modconv_f16_fused_torgb(simlib, golden, shape):
    """ToRGB's 1x1 modulated conv in the conv epilogue (hf_modconv3x3_f16_rgb_f32) + the finishing
    pass (bias + upsampled skip through hf_torgb_f32 with identity weights) == conv then ToRGB."""
    B, cin, cout, H, W = shape
    torch.manual_seed(9)
    x = torch.randn(B, cin, H, W)
    wgt = torch.randn(1, cout, cin, 3, 3)
    mw, mb, sty = torch.randn(cin, 16), torch.randn(cin), torch.randn(B, 16)
    nz, nw, bias = torch.randn(B, 1, H, W), torch.tensor([0.3]), torch.randn(cout)
    wrgb, mwr, mbr, styr = torch.randn(1, 3, cout, 1, 1), torch.randn(cout, 16), torch.randn(cout), torch.randn(B, 16)
    brgb, skip = torch.randn(3), torch.randn(B, 3, H // 2, W // 2)
    wt, wsq = M.prepare_weights(simlib, None, wgt)
    s = M.modulation(simlib, None, sty, mw, mb)
    dm = M.demod(simlib, None, s, wsq)
    hi, lo = M.split_weights_f16(simlib, None, wt)
    wtr, _ = M.prepare_weights(simlib, None, wrgb)
    sr = M.modulation(simlib, None, styr, mwr, mbr)
    k4 = O.blur_kernel_1d_to_2d(gain=4.0)
    assert M.modconv3x3_f16_supported(cin, cout, H, W) and cout % 32 == 0  # what the kernel takes (torgb_fusable: the policy)
    y, raw = M.modconv3x3_f16(simlib, None, x, hi, lo, 3, s, dm, nz, nw, bias, rgb=(wtr, sr))
    y_plain = M.modconv3x3_f16(simlib, None, x, hi, lo, 3, s, dm, nz, nw, bias)
    assert torch.equal(y, y_plain)
    slabs = M.torgb_slabs(cout)  # one partial sum per 64 (32) output channels, added by the finishing pass
    assert raw.shape == (B, 3 * slabs, H, W)
    rgb = M.torgb(simlib, None, raw, torch.eye(3).repeat(slabs, 1).reshape(1, 3 * slabs, 3), None, brgb, skip, k4)
    ref = M.torgb(simlib, None, y, wtr, sr, brgb, skip, k4)
    assert maxdiff(rgb, ref) < 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("shape", [(2, 16, 6, 5), (1, 8, 9, 40), (1, 24, 33, 70)])
def test_blur_split_output_is_the_split_of_the_fp32_blur(simlib, shape):
    """hf_blur_noise_bias_act_split_f16 == hf_blur_noise_bias_act_f32 followed by *s, fp16 (hi, lo)
    split and K-blocking, bit for bit (odd sizes, edges, strips shorter than the plane)."""
    B, C, h, w = shape
    torch.manual_seed(13)
    pitch = simlib.hf_modconv_up_pitch(w)
    tmp = torch.randn(B, C, 2 * h + 1, pitch)
    k4 = O.blur_kernel_1d_to_2d(gain=4.0)
    nz, nw, bias, s_next = torch.randn(B, 1, 2 * h, 2 * w), torch.tensor([0.3]), torch.randn(C), torch.rand(B, C) + 0.5
    out = torch.empty(B, C, 2 * h, 2 * w)
    M.check(simlib, simlib.hf_blur_noise_bias_act_f32(M._p(out), M._p(tmp), M._p(k4), M._p(nz), M._p(nw), 4 * h * w, M._p(bias),
                                                      B, C, 2 * h + 1, 2 * w + 1, pitch, 0.2, 2 ** 0.5, None), "blur")
    hi = torch.empty(B, C // 8, 2 * h, 2 * w, 8, dtype=torch.float16)
    lo = torch.empty_like(hi)
    M.check(simlib, simlib.hf_blur_noise_bias_act_split_f16(M._p(hi), M._p(lo), M._p(tmp), M._p(k4), M._p(nz), M._p(nw), 4 * h * w,
                                                            M._p(bias), M._p(s_next), B, C, 2 * h + 1, 2 * w + 1, pitch, 0.2,
                                                            2 ** 0.5, None), "blur split")
    v = out * s_next[:, :, None, None]
    eh = v.half()
    el = (v - eh.float()).half()
    blocked = lambda t: t.reshape(B, C // 8, 8, 2 * h, 2 * w).permute(0, 1, 3, 4, 2).contiguous()  # noqa: E731
    assert torch.equal(hi, blocked(eh))
    assert torch.equal(lo, blocked(el))


@pytest.mark.parametrize("nterms", [3, 1])
@pytest.mark.parametrize("cfg,shape", [(51, (2, 32, 64, 8, 32)), (51, (1, 48, 128, 20, 70)), (52, (1, 48, 64, 20, 40)),
                                       (53, (1, 16, 32, 16, 64))])
def test_modconv_f16_presplit_input(simlib, nterms, cfg, shape):
    """Pre-split, K-blocked activations staged by LDS-DMA (hf_modconv3x3_f16_pre_f32) give the
    bit-identical result of the kernel that loads fp32 activations and splits them itself:
    ragged planes (zero-filled halo), odd chunk counts, every tile shape."""
    B, cin, cout, H, W = shape
    torch.manual_seed(17)
    x = torch.randn(B, cin, H, W)
    wgt = torch.randn(1, cout, cin, 3, 3)
    s, dm = torch.rand(B, cin) + 0.5, torch.rand(B, cout) + 0.5
    nz, nw, bias = torch.randn(B, 1, H, W), torch.tensor([0.3]), torch.randn(cout)
    wt, _ = M.prepare_weights(simlib, None, wgt)
    hi, lo = M.split_weights_f16(simlib, None, wt)
    xh, xl = M.split_activation_reference(x, s)
    act = M.SplitActivation(xh, xl, None)
    try:
        simlib.hf_debug_set_dispatch(cfg if cfg in (51, 52) else 0, 0)
        ref = M.modconv3x3_f16(simlib, None, x, hi, lo, nterms, s, dm, nz, nw, bias)
        y = M.modconv3x3_f16_pre(simlib, None, act, hi, lo, nterms, dm, nz, nw, bias)
        assert simlib.hf_debug_last_path() == 520 + cfg
    finally:
        simlib.hf_debug_set_dispatch(0, 0)
    assert torch.equal(y, ref)


@pytest.mark.parametrize("shape", [(2, 64, 64, 8, 32), (1, 32, 32, 16, 32), (2, 64, 32, 20, 40)])
def test_presplit_chain_same_res_to_transposed(simlib, shape):
    """Split output of the same-resolution epilogue (hf_modconv3x3_f16_pre_f32 split_hi/lo) is the
    split of s_next * out bit for bit, and the transposed conv on it (hf_modconv3x3_up_f16_pre_f32,
    incl. rim tiles) equals the transposed conv that loads the fp32 activation."""
    B, cin, cup, H, W = shape
    torch.manual_seed(23)
    x = torch.randn(B, cin, H, W)
    w1, w2 = torch.randn(1, cin, cin, 3, 3), torch.randn(1, cup, cin, 3, 3)
    s1, d1 = torch.rand(B, cin) + 0.5, torch.rand(B, cin) + 0.5
    s2, d2 = torch.rand(B, cin) + 0.5, torch.rand(B, cup) + 0.5
    nz, nw, bias = torch.randn(B, 1, H, W), torch.tensor([0.3]), torch.randn(cin)
    wt1, _ = M.prepare_weights(simlib, None, w1)
    wt2, _ = M.prepare_weights(simlib, None, w2)
    h1 = M.split_weights_f16(simlib, None, wt1)
    h2 = M.split_weights_f16(simlib, None, wt2)
    act = M.SplitActivation(*M.split_activation_reference(x, s1), None)
    if not M.modconv3x3_f16_supported(cin, cin, H, W):
        pytest.skip("shape not taken by the same-resolution fp16 kernel")
    out, nxt = M.modconv3x3_f16_pre(simlib, None, act, h1[0], h1[1], 3, d1, nz, nw, bias, split_for=s2)
    eh, el = M.split_activation_reference(out, s2)
    assert torch.equal(nxt.hi, eh) and torch.equal(nxt.lo, el)
    only_split = M.modconv3x3_f16_pre(simlib, None, act, h1[0], h1[1], 3, d1, nz, nw, bias, split_for=s2, want_out=False)
    assert only_split[0] is None and torch.equal(only_split[1].hi, eh)
    k4 = O.blur_kernel_1d_to_2d(gain=4.0)
    nz2, b2 = torch.randn(B, 1, 2 * H, 2 * W), torch.randn(cup)
    ref = M.modconv3x3_up(simlib, None, out, wt2, s2, d2, k4, nz2, nw, b2, f16=(h2[0], h2[1], 3))
    y = M.modconv3x3_up(simlib, None, nxt, wt2, None, d2, k4, nz2, nw, b2, f16=(h2[0], h2[1], 3))
    assert simlib.hf_debug_last_path() in (581, 583)
    assert torch.equal(y, ref)


def test_style_batch_equals_per_layer_launches(simlib):
    """hf_style_batch_f32 (every layer's modulation + demodulation in two launches) against
    hf_modulation_f32 / hf_demod_f32 per layer: identical values, strided W+ rows, a job without
    demodulation (ToRGB), different channel counts per job."""
    import types

    torch.manual_seed(29)
    B, sd = 3, 48
    latent = torch.randn(B, 7, sd)
    convs, rows = [], []
    for cin, cout, demod, row in [(16, 24, True, 0), (40, 3, False, 5), (8, 8, True, 2)]:
        w = torch.randn(1, cout, cin, 3 if demod else 1, 3 if demod else 1)
        wt, wsq = M.prepare_weights(simlib, None, w)
        c = types.SimpleNamespace(in_channel=cin, out_channel=cout, demodulate=demod,
                                  modulation=types.SimpleNamespace(weight=torch.randn(cin, sd), bias=torch.randn(cin)),
                                  prepared=lambda wt=wt, wsq=wsq: (wt, wsq))
        convs.append(c)
        rows.append(row)
    table, layout, total = M.style_job_table(convs, rows, B, "cpu")
    res = M.style_batch(simlib, None, latent, table, layout, total, 40, 24)
    for c, row, (s, d) in zip(convs, rows, res):
        s_ref = M.modulation(simlib, None, latent[:, row], c.modulation.weight, c.modulation.bias)
        if c.demodulate:
            d_ref = M.demod(simlib, None, s_ref, c.prepared()[1])
            # the batch applies hf_style_normalize_f32: per sample s * 2^-e, d * 2^e (exact), max|s| in [1, 2)
            e = torch.floor(torch.log2(s_ref.abs().amax(1, keepdim=True)))
            assert torch.equal(s, s_ref * torch.exp2(-e)) and torch.equal(d, d_ref * torch.exp2(e))
            assert float(s.abs().amax(1).min()) >= 1.0 and float(s.abs().amax(1).max()) < 2.0
            M.style_normalize(simlib, None, s_ref, d_ref)  # the per-layer entry point does the same, in place
            assert torch.equal(s, s_ref) and torch.equal(d, d_ref)
        else:
            assert torch.equal(s, s_ref)  # no demodulation (ToRGB): nothing could absorb the factor
            assert d is None


def test_f16_split_range(simlib):
    """The fp16 (hi, lo) operand split is range-safe: a style 1e5 times larger than usual (normalised
    away, exactly), activations beyond the fp16 range (both parts saturate: representable up to
    131008, no inf/NaN) and a clamp counter for what lies beyond; tiny activations keep fp32-class
    accuracy because weights and styles are pre-scaled into the normal fp16 range."""
    B, cin, cout, H, W = 1, 32, 64, 8, 32
    torch.manual_seed(3)
    wgt = torch.randn(1, cout, cin, 3, 3)
    mw, mb, sty = torch.randn(cin, 16), torch.randn(cin), torch.randn(B, 16)
    wt, wsq = M.prepare_weights(simlib, None, wgt)
    hi, lo = M.split_weights_f16(simlib, None, wt)
    x = torch.randn(B, cin, H, W)
    for sty_scale, x_scale in [(1e5, 1.0), (1e-5, 1.0), (1.0, 5e3), (1.0, 1e-2), (1.0, 1e-4), (1e4, 5e3)]:
        simlib.hf_f16_overflow_count(1)
        s = M.modulation(simlib, None, sty * sty_scale, mw * 1.0, mb * sty_scale)
        dm = M.demod(simlib, None, s, wsq)
        M.style_normalize(simlib, None, s, dm)
        xs = x * x_scale
        ref = M.modconv3x3(simlib, None, xs, wt, s, dm, None, None, None)
        y = M.modconv3x3_f16(simlib, None, xs, hi, lo, 3, s, dm, None, None, None)
        assert torch.isfinite(y).all()
        # error model (include/hairfast_hip.h): 2^-22 relative per operand while |s*x| >= 2^-3, an absolute
        # 2^-25 below that - i.e. relative to the output ~3e-8 / rms(s*x) once the activations are tiny
        tol = 5e-6 + 6e-8 / min(1.0, x_scale)
        assert maxdiff(y, ref) < tol * float(ref.abs().max()), (sty_scale, x_scale)
        assert M.f16_overflow_count(simlib) == 0
    # beyond 2 * 65504 after normalisation (max|s| in [1, 2), |x| up to ~4 * 4e4): clamped, finite, counted
    s = M.modulation(simlib, None, sty, mw, mb)
    dm = M.demod(simlib, None, s, wsq)
    M.style_normalize(simlib, None, s, dm)
    y = M.modconv3x3_f16(simlib, None, x * 4e4, hi, lo, 3, s, dm, None, None, None)
    assert torch.isfinite(y).all()
    assert M.f16_overflow_count(simlib, reset=True) > 0
    assert M.f16_overflow_count(simlib) == 0


@pytest.mark.parametrize("shape,use_skip", [((2, 64, 4, 4), False), ((3, 136, 8, 12), True), ((1, 512, 16, 16), True)])
def test_torgb_small_plane_kernel(simlib, shape, use_skip):
    """hf_torgb_f32's small-plane kernel (channels split over 16 waves, LDS reduction) against a
    direct torch statement of ToRGB (model.py:356-365): ragged pixel counts, cin not a multiple of 16."""
    B, cin, H, W = shape
    torch.manual_seed(31)
    x, wt, s, bias = torch.randn(B, cin, H, W), torch.randn(1, cin, 3), torch.rand(B, cin) + 0.5, torch.randn(3)
    skip = torch.randn(B, 3, H // 2, W // 2) if use_skip else None
    k4 = O.blur_kernel_1d_to_2d(gain=4.0)
    y = M.torgb(simlib, None, x, wt, s, bias, skip, k4 if use_skip else None)
    ref = torch.einsum("bihw,ic,bi->bchw", x, wt[0], s) + bias.view(1, 3, 1, 1)
    if use_skip:
        ref = ref + O.upfirdn2d(skip, k4, up=2, pad=(2, 1))
    assert maxdiff(y, ref) < 1e-4 * max(1.0, float(ref.abs().max()))


def test_mapping_network_kernels(simlib, golden):
    """Row a11: PixelNorm + 8 x EqualLinear(lr_mul 0.01, fused_lrelu) through hf_pixel_norm_f32 /
    hf_equal_linear_f32 against the reference's golden w (and the oracle), 3 and 11 rows (> 8: two launches)."""
    size, cm, n_mlp, _, _ = C.GENERATOR_CASES["g64"]
    P = C.generator_params(O.generator_param_shapes(size, 512, n_mlp, cm))
    for B in (3, 11):
        z = C.mapping_inputs(B)
        x = M.pixel_norm(simlib, None, z)
        for i in range(1, n_mlp + 1):
            x = M.equal_linear(simlib, None, x, P[f"style.{i}.weight"], P[f"style.{i}.bias"], 0.01, True)
        ref = O.mapping_network(P, z, n_mlp=n_mlp)
        assert maxdiff(x, ref) < 1e-5 * max(1.0, float(ref.abs().max()))
        if B == 3:
            assert maxdiff(x, torch.from_numpy(golden("generator_64.npz")["g64_mapping_w"])) < 1e-5
    # no activation, no bias, strided input rows
    xx = torch.randn(5, 40)[:, :24]
    w = torch.randn(7, 24)
    y = M.equal_linear(simlib, None, xx, w, None, 1.0, False)
    assert maxdiff(y, xx @ w.t() / 24 ** 0.5) < 1e-5


@pytest.mark.parametrize("B,cin,cout,H,W", [(1, 16, 32, 16, 32), (2, 32, 32, 20, 45), (1, 16, 64, 30, 61)])
def test_modconv_up_fused_blur(simlib, B, cin, cout, H, W):
    """csrc/convh.hip FUSE: transposed conv + 4x4 blur + noise + bias + lrelu in one kernel (overlapping tiles of
    16 x 32 phase positions, horizontal taps by lane shifts, vertical taps through the free LDS stage buffer)
    against the two-pass path and the oracle: fp32 output, split output, pre-split input, several tiles per
    dimension with ragged edges, two cout tiles."""
    torch.manual_seed(B + W)
    x = torch.randn(B, cin, H, W)
    wgt = torch.randn(1, cout, cin, 3, 3)
    mw, mb, sty = torch.randn(cin, 16), torch.randn(cin), torch.randn(B, 16)
    nz, nw, bias = torch.randn(B, 1, 2 * H, 2 * W), torch.tensor([0.3]), torch.randn(cout)
    wt, wsq = M.prepare_weights(simlib, None, wgt)
    s = M.modulation(simlib, None, sty, mw, mb)
    dm = M.demod(simlib, None, s, wsq)
    M.style_normalize(simlib, None, s, dm)
    hi, lo = M.split_weights_f16(simlib, None, wt)
    k4 = O.blur_kernel_1d_to_2d(gain=4.0)
    fac = M.blur_factors(k4)
    assert fac is not None and M.blur_factors(torch.rand(4, 4)) is None
    assert M.modconv3x3_up_fused_supported(cin, cout, H, W)
    ref = M.modconv3x3_up(simlib, None, x, wt, s, dm, k4, nz, nw, bias, f16=(hi, lo, 3))
    y = M.modconv3x3_up_fused(simlib, None, x, hi, lo, s, dm, fac, nz, nw, bias)
    assert simlib.hf_debug_last_path() == 573
    scale = max(1.0, float(ref.abs().max()))
    assert maxdiff(y, ref) < 2e-6 * scale  # same products; the separable 4+4-tap blur reassociates the 16-tap sum
    full = O.fused_leaky_relu(O.modulated_conv2d(x, sty, wgt, mw, mb, True, True) + nw * nz, bias)
    assert maxdiff(y, full) < TOL * max(1.0, float(full.abs().max()))
    # split output for the next conv + pre-split input
    s2 = torch.rand(B, cout) + 0.5
    sp = M.modconv3x3_up_fused(simlib, None, x, hi, lo, s, dm, fac, nz, nw, bias, split_for=s2)
    eh, el = M.split_activation_reference(y, s2)
    assert torch.equal(sp.hi, eh) and torch.equal(sp.lo, el)
    xh, xl = M.split_activation_reference(x, s)
    y3 = M.modconv3x3_up_fused(simlib, None, M.SplitActivation(xh, xl, None), hi, lo, None, dm, fac, nz, nw, bias, split_for=s2)
    assert simlib.hf_debug_last_path() == 593
    assert torch.equal(y3.hi, eh) and torch.equal(y3.lo, el)
    y6 = M.modconv3x3_up_fused(simlib, None, M.SplitActivation(xh, xl, None), hi, lo, None, dm, fac, nz, nw, bias)  # fp32 output
    assert torch.equal(y6, y)


def test_modconv_up_fused_blur_plain_fp16_operands(simlib):
    """The one-kernel upsampling StyledConv with plain fp16 operands (nterms 1, BASELINE.json configs[4]; wt_lo NULL at the
    C ABI): same products as the two-pass fp16 path (operands rounded identically; the separable blur reassociates), the
    split output has a hi part only = fp16(s_next * out), the stage buffer is sized for the epilogue's exchange."""
    B, cin, cout, H, W = 2, 32, 32, 20, 45
    torch.manual_seed(3)
    x = torch.randn(B, cin, H, W)
    wgt = torch.randn(1, cout, cin, 3, 3)
    mw, mb, sty = torch.randn(cin, 16), torch.randn(cin), torch.randn(B, 16)
    nz, nw, bias = torch.randn(B, 1, 2 * H, 2 * W), torch.tensor([0.3]), torch.randn(cout)
    wt, wsq = M.prepare_weights(simlib, None, wgt)
    s = M.modulation(simlib, None, sty, mw, mb)
    dm = M.demod(simlib, None, s, wsq)
    M.style_normalize(simlib, None, s, dm)
    hi, lo = M.split_weights_f16(simlib, None, wt)
    k4 = O.blur_kernel_1d_to_2d(gain=4.0)
    fac = M.blur_factors(k4)
    ref = M.modconv3x3_up(simlib, None, x, wt, s, dm, k4, nz, nw, bias, f16=(hi, lo, 1))
    y = M.modconv3x3_up_fused(simlib, None, x, hi, None, s, dm, fac, nz, nw, bias, nterms=1)
    scale = max(1.0, float(ref.abs().max()))
    assert maxdiff(y, ref) < 2e-6 * scale
    full = O.fused_leaky_relu(O.modulated_conv2d(x, sty, wgt, mw, mb, True, True) + nw * nz, bias)
    assert maxdiff(y, full) < 2e-2 * max(1.0, float(full.abs().max()))  # fp16 operands
    s2 = torch.rand(B, cout) + 0.5
    xh, _ = M.split_activation_reference(x, s)
    sp = M.modconv3x3_up_fused(simlib, None, M.SplitActivation(xh, None, None), hi, None, None, dm, fac, nz, nw, bias, split_for=s2, nterms=1)
    assert sp.lo is None
    y2 = M.modconv3x3_up_fused(simlib, None, M.SplitActivation(xh, None, None), hi, None, None, dm, fac, nz, nw, bias, nterms=1)
    eh, _ = M.split_activation_reference(y2, s2)
    assert torch.equal(sp.hi, eh)


@pytest.mark.parametrize("blocks", [1, 3])
def test_modconv_up_fused_blur_persistent_walk(simlib, blocks):
    """The fused kernel under the resident-block tile walk: the LDS stage buffer that carries the vertical
    exchange is handed back to the next tile's DMA prefetch; several images (second parameter slot)."""
    B, cin, cout, H, W = 3, 32, 32, 30, 40
    torch.manual_seed(77)
    x = torch.randn(B, cin, H, W)
    wgt = torch.randn(1, cout, cin, 3, 3)
    mw, mb, sty = torch.randn(cin, 16), torch.randn(cin), torch.randn(B, 16)
    nz, nw, bias = torch.randn(B, 1, 2 * H, 2 * W), torch.tensor([0.3]), torch.randn(cout)
    wt, wsq = M.prepare_weights(simlib, None, wgt)
    s = M.modulation(simlib, None, sty, mw, mb)
    dm = M.demod(simlib, None, s, wsq)
    hi, lo = M.split_weights_f16(simlib, None, wt)
    k4 = O.blur_kernel_1d_to_2d(gain=4.0)
    ref = M.modconv3x3_up_fused(simlib, None, x, hi, lo, s, dm, M.blur_factors(k4), nz, nw, bias)
    try:
        simlib.hf_debug_set_persistent_blocks(blocks)
        y = M.modconv3x3_up_fused(simlib, None, x, hi, lo, s, dm, M.blur_factors(k4), nz, nw, bias)
        xh, xl = M.split_activation_reference(x, s)
        y2 = M.modconv3x3_up_fused(simlib, None, M.SplitActivation(xh, xl, None), hi, lo, None, dm, M.blur_factors(k4), nz, nw, bias)
    finally:
        simlib.hf_debug_set_persistent_blocks(0)
    assert torch.equal(y, ref) and torch.equal(y2, ref)
    full = O.fused_leaky_relu(O.modulated_conv2d(x, sty, wgt, mw, mb, True, True) + nw * nz, bias)
    assert maxdiff(y, full) < TOL * max(1.0, float(full.abs().max()))


@pytest.mark.parametrize("B,H,W", [(1, 16, 64), (2, 40, 128), (1, 64, 192)])
def test_conv_rows_pipeline_equals_tiled_kernel(simlib, B, H, W):
    """csrc/convrow.hip (32 -> 32 channels, pre-split input: the generator's 1024^2 conv as a row pipeline with the weights in
    registers and an 18-row LDS ring) against the tiled kernel it replaces (hf_debug_set_tuning bit 4), bit for bit: fp32
    output, fused ToRGB partial sums, noise; strips at both image borders, several super-steps, ring wrap-around."""
    torch.manual_seed(H + W)
    cin = cout = 32
    x = torch.randn(B, cin, H, W)
    wgt = torch.randn(1, cout, cin, 3, 3)
    s, dm = torch.rand(B, cin) + 0.5, torch.rand(B, cout) + 0.5
    nz, nw, bias = torch.randn(B, 1, H, W), torch.tensor([0.3]), torch.randn(cout)
    rgb_w, rgb_s = torch.randn(cout, 3) * 0.2, torch.rand(B, cout) + 0.5
    wt, _ = M.prepare_weights(simlib, None, wgt)
    hi, lo = M.split_weights_f16(simlib, None, wt)
    xh, xl = M.split_activation_reference(x, s)
    act = M.SplitActivation(xh, xl, None)
    try:
        simlib.hf_debug_set_tuning(16)
        ref_out, ref_raw = M.modconv3x3_f16_pre(simlib, None, act, hi, lo, 3, dm, nz, nw, bias, rgb=(rgb_w, rgb_s))
        assert simlib.hf_debug_last_path() in (573, 575)
    finally:
        simlib.hf_debug_set_tuning(0)
    out, raw = M.modconv3x3_f16_pre(simlib, None, act, hi, lo, 3, dm, nz, nw, bias, rgb=(rgb_w, rgb_s))
    assert simlib.hf_debug_last_path() == 579
    assert torch.equal(out, ref_out) and torch.equal(raw, ref_raw)
    only_raw = M.modconv3x3_f16_pre(simlib, None, act, hi, lo, 3, dm, nz, nw, bias, rgb=(rgb_w, rgb_s), want_out=False)[1]
    assert torch.equal(only_raw, ref_raw)
    # round 6: the epilogue finishes ToRGB (bias + x2-upsampled skip through a 16-row LDS ring of skip rows) - against the raw
    # product + the finishing launch (ToRGB.finish: hf_torgb_f32 with identity weights), bit for bit; image borders, ring wrap
    skip, rgb_bias, k4 = torch.randn(B, 3, H // 2, W // 2), torch.randn(1, 3, 1, 1), O.blur_kernel_1d_to_2d(gain=4.0)
    want_img = M.torgb(simlib, None, ref_raw, torch.eye(3).reshape(1, 3, 3), None, rgb_bias, skip, k4)
    img = M.modconv3x3_f16_pre_image(simlib, None, act, hi, lo, 3, dm, nz, nw, bias, (rgb_w, rgb_s), rgb_bias, skip, k4)
    assert simlib.hf_debug_last_path() == 579
    assert torch.equal(img, want_img)
    try:
        simlib.hf_debug_set_tuning(16)  # the tiled form asked for: the library declines, the caller keeps the two launches
        assert M.modconv3x3_f16_pre_image(simlib, None, act, hi, lo, 3, dm, nz, nw, bias, (rgb_w, rgb_s), rgb_bias, skip, k4) is None
    finally:
        simlib.hf_debug_set_tuning(0)
    plain = M.modconv3x3_f16_pre(simlib, None, act, hi, lo, 3, dm, None, None, bias)
    assert simlib.hf_debug_last_path() == 579
    want = M.modconv3x3_f16(simlib, None, x, hi, lo, 3, s, dm, None, None, bias)
    assert torch.equal(plain, want)
    # plain fp16 operands (nterms 1, BASELINE.json configs[4]): hi parts only - the same row pipeline, bit-equal to the tiled form
    act1 = M.SplitActivation(xh, None, None)
    try:
        simlib.hf_debug_set_tuning(16)
        ref1_out, ref1_raw = M.modconv3x3_f16_pre(simlib, None, act1, hi, lo, 1, dm, nz, nw, bias, rgb=(rgb_w, rgb_s))
        assert simlib.hf_debug_last_path() in (573, 575)
    finally:
        simlib.hf_debug_set_tuning(0)
    out1, raw1 = M.modconv3x3_f16_pre(simlib, None, act1, hi, lo, 1, dm, nz, nw, bias, rgb=(rgb_w, rgb_s))
    assert simlib.hf_debug_last_path() == 579
    img1 = M.modconv3x3_f16_pre_image(simlib, None, act1, hi, lo, 1, dm, nz, nw, bias, (rgb_w, rgb_s), rgb_bias, skip, k4)
    assert torch.equal(img1, M.torgb(simlib, None, ref1_raw, torch.eye(3).reshape(1, 3, 3), None, rgb_bias, skip, k4))
    assert torch.equal(out1, ref1_out) and torch.equal(raw1, ref1_raw) and not torch.equal(out1, out)


@pytest.mark.parametrize("nterms", [3, 1])
def test_modconv_f16_tile_forms_51_and_52_give_equal_bits(simlib, nterms):
    """The two same-resolution tile forms of csrc/convh.hip (51: 64 co x 256 px, 2 x 4 waves of 1 x 2 MFMA tiles; 52: 64 co x
    512 px, 8 waves of 2 x 2 tiles) walk K in the same order - chunk, tap, (hi*hi, hi*lo, lo*hi) - so a sample's bits do
    not depend on which one a launch takes: the choice follows the whole launch also under batch-invariant plans.  Ragged
    plane, odd chunk count, fp32 and pre-split input, noise + bias + lrelu epilogue."""
    B, cin, cout, H, W = 2, 48, 64, 36, 40
    torch.manual_seed(23)
    x = torch.randn(B, cin, H, W)
    wgt = torch.randn(1, cout, cin, 3, 3)
    s, dm = torch.rand(B, cin) + 0.5, torch.rand(B, cout) + 0.5
    nz, nw, bias = torch.randn(B, 1, H, W), torch.tensor([0.3]), torch.randn(cout)
    wt, _ = M.prepare_weights(simlib, None, wgt)
    hi, lo = M.split_weights_f16(simlib, None, wt)
    act = M.SplitActivation(*M.split_activation_reference(x, s), None)
    out = {}
    try:
        for cfg in (51, 52):
            simlib.hf_debug_set_dispatch(cfg, 0)
            out[cfg] = M.modconv3x3_f16(simlib, None, x, hi, lo, nterms, s, dm, nz, nw, bias)
            assert simlib.hf_debug_last_path() == 500 + cfg
            out[cfg, "pre"] = M.modconv3x3_f16_pre(simlib, None, act, hi, lo, nterms, dm, nz, nw, bias)
            assert simlib.hf_debug_last_path() == 520 + cfg
    finally:
        simlib.hf_debug_set_dispatch(0, 0)
    assert torch.equal(out[51], out[52]) and torch.equal(out[51, "pre"], out[52, "pre"]) and torch.equal(out[51], out[51, "pre"])


def test_small_plane_tap_gemm_virtual_split_k(simlib):
    """hf_modconv3x3_small_f16_f32 (the tap GEMM of the generator's small planes + small_combine) under batch-invariant
    plans: a batch that fills "the chip" keeps the canonical K partition inside its blocks (one slab instead of `splits`) -
    the bits of the one-sample launches."""
    torch.manual_seed(29)
    B, cin, cout, H, W = 4, 256, 64, 8, 8
    x = torch.randn(B, cin, H, W)
    wgt = torch.randn(1, cout, cin, 3, 3)
    s, dm = torch.rand(B, cin) + 0.5, torch.rand(B, cout) + 0.5
    nz, nw, bias = torch.randn(B, 1, H, W), torch.tensor([0.3]), torch.randn(cout)
    wt, _ = M.prepare_weights(simlib, None, wgt)
    w9 = M.split_weights_small(simlib, None, wt)
    prev = simlib.hf_set_batch_invariant(1)
    try:
        ref = torch.cat([M.modconv3x3_small(simlib, None, x[b:b + 1], w9, 3, s[b:b + 1], dm[b:b + 1], nz[b:b + 1], nw, bias, cout)
                         for b in range(B)])
        n1 = simlib.hf_modconv3x3_small_workspace_floats(B, cin, cout, H, W)
        # canonical batch 3: 2 pixel tiles (two 8 x 8 images each) x 9 channel tiles = 18 blocks -> splits; batch 4: 18 too...
        simlib.hf_debug_set_tuning(18 << 24)
        y = M.modconv3x3_small(simlib, None, x, w9, 3, s, dm, nz, nw, bias, cout)
        n2 = simlib.hf_modconv3x3_small_workspace_floats(B, cin, cout, H, W)
    finally:
        simlib.hf_debug_set_tuning(0)
        simlib.hf_set_batch_invariant(prev)
    assert torch.equal(y, ref)
    assert n2 < n1  # one slab instead of `splits`: the virtual form ran


@pytest.mark.parametrize("up", [False, True])
def test_modconv_f16_block_order_does_not_change_results(simlib, up):
    """csrc/convh.hip launches (tile walkers, cout tiles) or - ConvParams::swap_xy, forced here by hf_debug_set_tuning bit 3 -
    (cout tiles, tile walkers): which of the input tile and the cout tile's weights stays in an XCD's L2.  Identical results:
    same-resolution conv with split output and the transposed conv, several cout tiles, a persistent tile walk."""
    B, cin, cout, H, W = 2, 32, 128, 16, 32
    torch.manual_seed(37)
    x = torch.randn(B, cin, H, W)
    wgt = torch.randn(1, cout, cin, 3, 3)
    s, dm, s2 = torch.rand(B, cin) + 0.5, torch.rand(B, cout) + 0.5, torch.rand(B, cout) + 0.5
    nz, nw, bias = torch.randn(B, 1, H, W), torch.tensor([0.3]), torch.randn(cout)
    wt, _ = M.prepare_weights(simlib, None, wgt)
    hi, lo = M.split_weights_f16(simlib, None, wt)
    act = M.SplitActivation(*M.split_activation_reference(x, s), None)
    nz2 = torch.randn(B, 1, 2 * H, 2 * W)
    outs = []
    try:
        for tune in (0, 8):
            simlib.hf_debug_set_tuning(tune)
            simlib.hf_debug_set_persistent_blocks(4)  # 4 resident blocks over 2 cout tiles: every block walks several tiles
            if up:
                outs.append(M.modconv3x3_up(simlib, None, act, wt, None, dm, O.blur_kernel_1d_to_2d(gain=4.0), nz2, nw, bias, f16=(hi, lo, 3)))
            else:
                y = M.modconv3x3_f16_pre(simlib, None, act, hi, lo, 3, dm, nz, nw, bias, split_for=s2)
                parts = y if isinstance(y, (tuple, list)) else (y,)
                flat = []
                for t_ in parts:
                    flat += [t_.hi.float().flatten(), t_.lo.float().flatten()] if isinstance(t_, M.SplitActivation) else ([t_.flatten()] if torch.is_tensor(t_) else [])
                outs.append(torch.cat(flat))
    finally:
        simlib.hf_debug_set_tuning(0)
        simlib.hf_debug_set_persistent_blocks(0)
    assert outs[0].numel() > 0 and torch.equal(outs[0], outs[1])
