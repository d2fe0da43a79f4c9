"""CPU tests (kernel sources interpreted by tests/hipsim) of the 1x1-conv GEMM on the fp16 matrix cores
(csrc/gemm_h.hip, hf_conv1x1_f16_f32): tile geometry (flat pixel tiles, several small images per tile), stride 2,
input affine, epilogue options, grouped launches, split-K, both operand modes."""
import pytest
import torch
import torch.nn.functional as F

from hairfastgan_amd import _marshal as M


def _prep(simlib, w):
    """torch conv weight [cout, cin, 1, 1] (or grouped [G, cout, cin, 1, 1]) -> (hi, lo) blobs."""
    if w.ndim == 5:
        wt = torch.stack([M.conv_prepare(simlib, None, wg) for wg in w])
    else:
        wt = M.conv_prepare(simlib, None, w)
    return M.conv_split_weights_f16(simlib, None, wt)


@pytest.mark.parametrize("nterms", [3, 1])
def test_conv1x1_forms(simlib, nterms):
    torch.manual_seed(0)
    tol = 2e-5 if nterms == 3 else 2e-2
    # plain + bias + PReLU + residual, odd plane (tile overhang), two images
    x = torch.randn(2, 64, 9, 11)
    w, b, sl, res = torch.randn(64, 64, 1, 1) * 0.1, torch.randn(64), torch.rand(64), torch.randn(2, 64, 9, 11)
    hi, lo = _prep(simlib, w)
    y = M.conv1x1_f16(simlib, None, x, hi, lo, nterms, 64, bias=b, act=M.ACT_PRELU, slope=sl, residual=res)
    ref = F.prelu(F.conv2d(x, w, b), sl) + res
    assert float((y - ref).abs().max()) < tol * max(1.0, float(ref.abs().max()))
    # stride 2 + input affine + output scale, residual before the activation
    x = torch.randn(1, 32, 14, 10)
    w = torch.randn(128, 32, 1, 1) * 0.2
    isc, ish, osc, b = torch.rand(32) + 0.5, torch.randn(32) * 0.1, torch.rand(128) + 0.5, torch.randn(128)
    res = torch.randn(1, 128, 7, 5)
    hi, lo = _prep(simlib, w)
    y = M.conv1x1_f16(simlib, None, x, hi, lo, nterms, 128, stride=2, in_scale=isc, in_shift=ish, out_scale=osc, bias=b,
                      act=M.ACT_LRELU | M.ACT_RESIDUAL_FIRST, alpha=0.2, residual=res)
    xin = x * isc[None, :, None, None] + ish[None, :, None, None]
    ref = F.leaky_relu(F.conv2d(xin, w, stride=2) * osc[None, :, None, None] + b[None, :, None, None] + res, 0.2)
    assert float((y - ref).abs().max()) < tol * max(1.0, float(ref.abs().max()))
    # small power-of-two planes: several whole images per tile (5 images of 4 x 4)
    x = torch.randn(5, 32, 4, 4)
    w = torch.randn(64, 32, 1, 1) * 0.2
    hi, lo = _prep(simlib, w)
    y = M.conv1x1_f16(simlib, None, x, hi, lo, nterms, 64)
    ref = F.conv2d(x, w)
    assert float((y - ref).abs().max()) < tol * max(1.0, float(ref.abs().max()))


def test_conv1x1_split_k_and_groups(simlib):
    torch.manual_seed(1)
    # feature-major Linear: [1, K, images, tokens] with K = 256 (8 stages) and few pixels -> split-K + reduce pass
    x = torch.randn(1, 256, 2, 17)
    w, b, res = torch.randn(64, 256, 1, 1) * 0.05, torch.randn(64), torch.randn(1, 64, 2, 17)
    assert simlib.hf_conv1x1_f16_workspace_floats(1, 256, 64, 2, 17, 1, 1) > 0
    hi, lo = _prep(simlib, w)
    y = M.conv1x1_f16(simlib, None, x, hi, lo, 3, 64, bias=b, residual=res)
    ref = F.conv2d(x, w, b) + res
    assert float((y - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    # grouped launch with per-group inputs of one pixel each (SEAN's nineteen fc_mu layers): [G, B, cin, 1, 1]
    G, B = 3, 2
    xg = torch.randn(G, B, 64, 1, 1)
    wg, bg = torch.randn(G, 64, 64, 1, 1) * 0.1, torch.randn(G, 64)
    hi, lo = _prep(simlib, wg)
    y = M.conv1x1_f16(simlib, None, xg, hi, lo, 3, 64, bias=bg, act=M.ACT_LRELU, alpha=0.0, groups=G, x_shared=False)
    for g in range(G):
        ref = F.relu(F.conv2d(xg[g], wg[g], bg[g]))
        assert float((y[g] - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    # grouped, shared input
    xs = torch.randn(2, 32, 6, 6)
    wg2 = torch.randn(2, 64, 32, 1, 1) * 0.1
    hi, lo = _prep(simlib, wg2)
    y = M.conv1x1_f16(simlib, None, xs, hi, lo, 3, 64, groups=2)
    for g in range(2):
        assert float((y[g] - F.conv2d(xs, wg2[g])).abs().max()) < 2e-5 * 3


def test_conv1x1_presplit_input_and_small_plane_modconv(simlib):
    """Pre-split input (hf_split_activation_f16 -> LDS-DMA staging) gives the register-staged GEMM's result bit for bit;
    the small-plane modulated 3x3 conv built on it (nine taps as one GEMM + combine pass) equals the fp32 kernels, same
    resolution and transposed, power-of-two and odd planes."""
    torch.manual_seed(2)
    x = torch.randn(2, 64, 6, 10)
    w, b = torch.randn(128, 64, 1, 1) * 0.1, torch.randn(128)
    isc, ish = torch.rand(64) + 0.5, torch.randn(64) * 0.1
    hi, lo = _prep(simlib, w)
    plain = M.conv1x1_f16(simlib, None, x, hi, lo, 3, 128, in_scale=isc, in_shift=ish, bias=b)
    xs = M.split_activation_f16(simlib, None, x, isc, ish)
    pre = M.conv1x1_f16(simlib, None, xs, hi, lo, 3, 128, bias=b)
    assert torch.equal(plain, pre)
    for (B, cin, cout, h, ww) in [(3, 32, 64, 4, 4), (2, 32, 64, 5, 7), (1, 32, 64, 16, 16)]:
        xx, wgt = torch.randn(B, cin, h, ww), torch.randn(1, cout, cin, 3, 3)
        s, d = torch.rand(B, cin) + 0.5, torch.rand(B, cout) + 0.5
        nz, nw, bias = torch.randn(B, 1, h, ww), torch.tensor([0.3]), torch.randn(cout)
        wt, _ = M.prepare_weights(simlib, None, wgt)
        w9 = M.split_weights_small(simlib, None, wt)
        y = M.modconv3x3_small(simlib, None, xx, w9, 3, s, d, nz, nw, bias, cout)
        ref = M.modconv3x3(simlib, None, xx, wt, s, d, nz, nw, bias)
        assert float((y - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
        # the unmodulated form encoders/_fused.py routes dense small-plane convs to: no s / d, alpha 1 = no activation
        plain = M.modconv3x3_small(simlib, None, xx, w9, 3, None, None, None, None, bias, cout, alpha=1.0, scale=1.0)
        ref = F.conv2d(xx, wt.reshape(3, 3, cin, cout).permute(3, 2, 0, 1), bias, padding=1)  # wt = prepared (scaled) weights
        assert float((plain - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
        act = M.modconv3x3_small(simlib, None, xx, w9, 3, None, None, None, None, bias, cout, alpha=0.01, scale=1.0)
        assert float((act - F.leaky_relu(ref, 0.01)).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
        t = M.modconv3x3_small(simlib, None, xx, w9, 3, s, d, None, None, None, cout, upsample=True)
        pitch = simlib.hf_modconv_up_pitch(ww)
        tr = torch.zeros(B, cout, 2 * h + 1, pitch)
        n = simlib.hf_modconv_workspace_floats(B, cin, cout, h, ww, 1)
        ws = torch.empty(max(n, 1))
        assert simlib.hf_modconv3x3_up_f32(tr.data_ptr(), xx.data_ptr(), wt.data_ptr(), s.data_ptr(), d.data_ptr(), B, cin, cout, h, ww,
                                           pitch, ws.data_ptr() if n > 0 else None, n, None) == 0
        assert float((t[..., :2 * ww + 1] - tr[..., :2 * ww + 1]).abs().max()) < 2e-5 * max(1.0, float(tr.abs().max()))


def test_split_k_in_kernel_reduction_equals_the_second_pass(simlib):
    """hf_set_splitk_counters: with a registered zeroed counter buffer the LAST block of every output tile adds the z slabs
    and runs the epilogue inside the split-K launch (csrc/conv_common.h splitk_arrive_last) - the same bits as the
    splitk_reduce pass (z order, one shared per-element tail), counters zero again after every launch, on the three kernel
    families that split K: the tiled fp16-core conv, the 1x1 GEMM, the fp32-MFMA conv (incl. noise / per-image d)."""
    torch.manual_seed(5)
    counters = torch.zeros(4096, dtype=torch.int32)

    def both(fn):
        simlib.hf_set_splitk_counters(None, 0)
        two_pass = fn()
        simlib.hf_set_splitk_counters(counters.data_ptr(), counters.numel())
        try:
            in_kernel = fn()
        finally:
            simlib.hf_set_splitk_counters(None, 0)
        assert int(counters.abs().sum()) == 0
        assert torch.equal(two_pass, in_kernel)
        return in_kernel

    # tiled fp16-core conv (hf_conv2d_f16_f32): 2 tiles x 16 stages -> split; BN affine, PReLU, residual in the tail
    B, cin, cout, H, W = 2, 256, 64, 16, 16
    assert simlib.hf_conv2d_f16_workspace_floats(B, cin, cout, H, W, 1, 1) > 0
    x = torch.randn(B, cin, H, W)
    w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    g, bsh, slope, res = torch.rand(cout) + 0.5, torch.randn(cout) * 0.2, torch.rand(cout) * 0.5, torch.randn(B, cout, H, W)
    hi, lo = M.conv_split_weights_f16(simlib, None, M.conv_prepare(simlib, None, w))
    y = both(lambda: M.conv2d_f16(simlib, None, x, hi, lo, 3, cout, 1, out_scale=g, bias=bsh, act=M.ACT_PRELU, slope=slope, residual=res))
    want = F.prelu(F.conv2d(x, w, padding=1) * g.view(1, -1, 1, 1) + bsh.view(1, -1, 1, 1), slope) + res
    assert float((y - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))
    # 1x1 GEMM (hf_conv1x1_f16_f32), ragged pixel count
    xg = torch.randn(1, 256, 2, 17)
    wg, bg, rg = torch.randn(64, 256, 1, 1) * 0.05, torch.randn(64), torch.randn(1, 64, 2, 17)
    assert simlib.hf_conv1x1_f16_workspace_floats(1, 256, 64, 2, 17, 1, 1) > 0
    hg, lg = _prep(simlib, wg)
    y = both(lambda: M.conv1x1_f16(simlib, None, xg, hg, lg, 3, 64, bias=bg, residual=rg))
    assert float((y - (F.conv2d(xg, wg, bg) + rg)).abs().max()) < 2e-5 * 4
    # fp32-MFMA modulated conv on a small plane (split-K of modconv.hip): per-image s / d, noise, bias, leaky ReLU
    xm = torch.randn(2, 64, 8, 8)
    wm = torch.randn(1, 64, 64, 3, 3)
    wt, wsq = M.prepare_weights(simlib, None, wm)
    s, d = torch.rand(2, 64) + 0.5, torch.rand(2, 64) + 0.5
    nz, nw, bias = torch.randn(1, 1, 8, 8), torch.tensor([0.3]), torch.randn(64)
    assert simlib.hf_modconv_workspace_floats(2, 64, 64, 8, 8, 0) > 0
    both(lambda: M.modconv3x3(simlib, None, xm, wt, s, d, nz, nw, bias, 0.2, 2 ** 0.5))


def test_small_plane_upsampling_styledconv_in_two_launches(simlib, monkeypatch):
    """hf_modconv3x3_small_up_blur_f16_f32 (round 6: tap GEMM + ONE combine / blur / tail kernel, the (2h+1)^2 intermediate in
    LDS) against the three-launch path - tap GEMM, small_combine, blur4x4_noise_bias_act / blur4x4_split8 - bit for bit: fp32
    output and split output (with and without a lo part), odd planes, with and without noise / bias."""
    from oracle import ref_stylegan2 as O

    torch.manual_seed(11)
    k4 = O.blur_kernel_1d_to_2d(gain=4.0)
    for (B, cin, cout, h, ww) in [(3, 32, 64, 4, 4), (2, 64, 64, 5, 7), (1, 32, 128, 16, 16)]:
        xx, wgt = torch.randn(B, cin, h, ww), torch.randn(1, cout, cin, 3, 3)
        s, d, s2 = torch.rand(B, cin) + 0.5, torch.rand(B, cout) + 0.5, torch.rand(B, cout) + 0.5
        nz, nw, bias = torch.randn(B, 1, 2 * h, 2 * ww), torch.tensor([0.3]), torch.randn(cout)
        wt, _ = M.prepare_weights(simlib, None, wgt)
        w9 = M.split_weights_small(simlib, None, wt)
        for noise, b_ in ((nz, bias), (None, None), (None, bias)):
            for split_for in (None, (None, s2, True), (None, s2, False)):
                if b_ is None and split_for is not None:
                    continue
                res = {}
                for fused in (False, True):
                    monkeypatch.setattr(M, "SMALL_UP_FUSED", fused)
                    res[fused] = M.modconv3x3_up(simlib, None, xx, wt, s, d, k4, noise, nw if noise is not None else None, b_,
                                                 split_for=split_for, small=(w9, 3))
                    assert simlib.hf_debug_last_path() == (705 if fused else 704)
                if split_for is None:
                    assert torch.equal(res[False], res[True])
                else:
                    assert torch.equal(res[False].hi, res[True].hi)
                    assert (res[False].lo is None) == (res[True].lo is None)
                    if res[True].lo is not None:
                        assert torch.equal(res[False].lo, res[True].lo)


def test_conv1x1_128_channel_blocks_equal_64_channel_blocks(simlib):
    """Round 6: the 128-channel block form of the GEMM (gemm1x1_h<..., CW = 4>: eight waves share one activation stage) is a tile
    FORM - every output element keeps its K order - so it must equal the 64-channel form bit for bit: register-staged and
    pre-split input, bias + PReLU + residual epilogue, the canonical (virtual) K partition of the batch-invariant mode, odd planes.
    hf_debug_set_tuning: bits 24-31 = block count from which a launch counts as chip-filling (lowered so that small shapes take
    the form), bit 1 = never the 128-channel form."""
    torch.manual_seed(21)
    fill1 = 1 << 24
    for (B, cin, cout, h, w) in [(2, 64, 256, 13, 21), (1, 128, 128, 18, 17)]:
        x = torch.randn(B, cin, h, w)
        wgt, b, sl, res = torch.randn(cout, cin, 1, 1) * 0.1, torch.randn(cout), torch.rand(cout), torch.randn(B, cout, h, w)
        hi, lo = _prep(simlib, wgt)
        xs = M.split_activation_f16(simlib, None, x)
        for binv in (0, 1):
            prev = simlib.hf_set_batch_invariant(binv)
            try:
                for xin in (x, xs):
                    out = {}
                    for never in (0, 2):
                        simlib.hf_debug_set_tuning(fill1 | never)
                        out[never] = M.conv1x1_f16(simlib, None, xin, hi, lo, 3, cout, bias=b, act=M.ACT_PRELU, slope=sl, residual=res)
                        assert simlib.hf_debug_last_path() == (702 if never else 706)
                    assert torch.equal(out[0], out[2])
                    ref = F.prelu(F.conv2d(x, wgt, b), sl) + res
                    assert float((out[0] - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
            finally:
                simlib.hf_set_batch_invariant(prev)
                simlib.hf_debug_set_tuning(0)


def test_small_plane_tap_gemm_128_pixel_tiles_equal_256_pixel_tiles(simlib):
    """Round 6: the small-plane tap GEMM trades its 256-pixel tiles for 128-pixel ones when the last round of the 256-pixel
    form would be under half full (the generator's 16^2 layers at batch 8: 576 blocks on 512 slots).  A tile FORM: every
    output element keeps its K order, so both forms must agree bit for bit - same resolution and transposed, with and without
    the batch-invariant K partition.  Bits 24-31 of hf_debug_set_tuning lower the chip-filling block count (slots = 2 x fill) so
    that a small shape takes the rule: 16^2, cout 64 -> 9 channel tiles on 8 slots, a last round of one block."""
    torch.manual_seed(22)
    fill4 = 4 << 24
    B, cin, cout, h, w = 1, 64, 64, 16, 16
    xx, wgt = torch.randn(B, cin, h, w), torch.randn(1, cout, cin, 3, 3)
    s, d = torch.rand(B, cin) + 0.5, torch.rand(B, cout) + 0.5
    nz, nw, bias = torch.randn(B, 1, h, w), torch.tensor([0.3]), torch.randn(cout)
    wt, _ = M.prepare_weights(simlib, None, wgt)
    w9 = M.split_weights_small(simlib, None, wt)
    for binv in (0, 1):
        prev = simlib.hf_set_batch_invariant(binv)
        try:
            same, up = {}, {}
            for never in (0, 2):
                simlib.hf_debug_set_tuning(fill4 | never)
                same[never] = M.modconv3x3_small(simlib, None, xx, w9, 3, s, d, nz, nw, bias, cout)
                up[never] = M.modconv3x3_small(simlib, None, xx, w9, 3, s, d, None, None, None, cout, upsample=True)
            assert torch.equal(same[0], same[2]) and torch.equal(up[0][..., :2 * w + 1], up[2][..., :2 * w + 1])  # (pitch padding: unwritten)
            ref = M.modconv3x3(simlib, None, xx, wt, s, d, nz, nw, bias)
            assert float((same[0] - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
        finally:
            simlib.hf_set_batch_invariant(prev)
            simlib.hf_debug_set_tuning(0)
