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
