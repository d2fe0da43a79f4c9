"""CPU tests (kernel sources interpreted by tests/hipsim) of the face-parsing path:
the oracle against the reference's golden masks, and the HIP-path BiSeNet mirror against the oracle on a small image."""
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from hairfastgan_amd import _marshal as M
from oracle import cases as C
from oracle import ref_bisenet as BS


def test_oracle_reproduces_reference_masks(golden):
    """oracle/ref_bisenet.py vs the golden vectors make_golden.py produced with the real BiSeNet class, its label
    permutation and the nearest resize: identical logits samples and identical mask indices."""
    G = golden("bisenet.npz")
    P = C.bisenet_params()
    x = C.bisenet_input("320x384")
    logits = BS.bisenet_logits(P, x)
    f = logits.reshape(-1)
    step = max(1, f.numel() // 2048)
    assert np.array_equal(f[::step][:2048].numpy(), G["logits_samples_320x384"])
    assert np.array_equal(BS.get_segmentation(P, x, resize=False)[0, 0].numpy().astype(np.uint8), G["mask_320x384"])
    assert np.array_equal(BS.get_segmentation(P, x, resize=True)[0, 0].numpy().astype(np.uint8), G["mask256_320x384"])
    assert len(np.unique(G["mask_512"])) >= 5  # the synthetic parameters give a map with several regions


@pytest.fixture()
def sim_parsing(simlib, monkeypatch):
    import hairfastgan_amd.encoders  # noqa: F401
    import hairfastgan_amd.face_parsing  # noqa: F401

    for n in ("hairfastgan_amd.encoders._fused", "hairfastgan_amd.face_parsing"):
        mod = sys.modules[n]
        monkeypatch.setattr(mod, "lib", lambda: simlib)
        monkeypatch.setattr(mod, "stream", lambda: None)
        monkeypatch.setattr(mod, "require_gpu", lambda *a: None)
    return simlib


def test_small_ops(simlib):
    torch.manual_seed(1)
    x = torch.randn(2, 3, 9, 12)
    assert torch.equal(M.maxpool3x3s2(simlib, None, x), F.max_pool2d(x, 3, 2, 1))
    assert torch.equal(M.upsample_nearest(simlib, None, x, 18, 24), F.interpolate(x, (18, 24), mode="nearest"))
    assert torch.equal(M.upsample_nearest(simlib, None, x, 13, 17), F.interpolate(x, (13, 17), mode="nearest"))
    lg, ap, ab = torch.randn(2, 3), torch.randn(2, 3, 9, 12), torch.randn(2, 3)
    ref = x * (torch.sigmoid(lg)[:, :, None, None] + 1.0) + ap + ab[:, :, None, None]
    assert float((M.gate(simlib, None, x, lg, ap, ab, 1.0) - ref).abs().max()) < 1e-6
    # 7x7 stride-2 conv (the ResNet stem) and the residual-before-activation epilogue
    w = torch.randn(8, 3, 7, 7) * 0.1
    xx = torch.randn(1, 3, 20, 28)
    wt = M.conv_prepare(simlib, None, w)
    y = M.conv2d(simlib, None, xx, wt, 7, 2, act=M.ACT_LRELU, alpha=0.0)
    assert float((y - F.relu(F.conv2d(xx, w, stride=2, padding=3))).abs().max()) < 2e-5
    w3 = torch.randn(8, 8, 3, 3) * 0.1
    res = torch.randn(1, 8, 10, 14)
    y2 = M.conv2d(simlib, None, y, M.conv_prepare(simlib, None, w3), 3, 1, act=M.ACT_LRELU | M.ACT_RESIDUAL_FIRST, alpha=0.0,
                  residual=res)
    assert float((y2 - F.relu(F.conv2d(y, w3, padding=1) + res)).abs().max()) < 2e-5
    # parsing tail: bilinear (align_corners) + argmax + remap + nearest resize == the torch composition, bit for bit
    logits = torch.randn(2, 19, 8, 10)
    remap = torch.tensor(BS.LABEL_REMAP, dtype=torch.int32)
    full = F.interpolate(logits, (40, 56), mode="bilinear", align_corners=True).argmax(1, keepdim=True)
    want = torch.tensor(BS.LABEL_REMAP)[full]
    assert torch.equal(M.parsing_mask(simlib, None, logits, remap, (40, 56), (40, 56)), want)
    want_r = F.interpolate(want.float(), (16, 16), mode="nearest").long()
    assert torch.equal(M.parsing_mask(simlib, None, logits, remap, (40, 56), (16, 16)), want_r)


def test_bisenet_small_image_vs_oracle(sim_parsing):
    """The whole BiSeNet mirror (ResNet18 stem / blocks with the residual before the ReLU, both attention refinement
    modules, feature fusion, output head, fused interpolation + argmax + remap) on a 64 x 64 image."""
    from hairfastgan_amd.face_parsing import BiSeNet, get_segmentation

    P = C.bisenet_params()
    net = BiSeNet(19).eval()
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == BS.bisenet_param_shapes()
    assert list(net.state_dict()) == list(BS.bisenet_param_shapes())
    net.load_state_dict(P)
    x = C.bisenet_input("320x384")[:, :, :64, :64].contiguous()
    low = net.logits_low(x)
    ref_full = BS.bisenet_logits(P, x)
    got_full = F.interpolate(low, (64, 64), mode="bilinear", align_corners=True)
    scale = float(ref_full.abs().max())
    assert float((got_full - ref_full).abs().max()) < 1e-4 * scale
    # BiSeNet.parse = logits_low + hf_parsing_mask_i64: one interpreted forward serves the three checks (the network costs
    # ~20 s on the kernel interpreter); get_segmentation itself runs through the C ABI in tests/test_gpu_parsing.py
    remap = net._prepared()["remap"]
    mask = M.parsing_mask(sim_parsing, None, low, remap, (64, 64), (64, 64))
    ref_mask = BS.get_segmentation(P, x, resize=False)
    top2 = ref_full[0].topk(2, dim=0).values
    margin = top2[0] - top2[1]
    flips = mask[0, 0] != ref_mask[0, 0]
    assert int(flips.sum()) == 0 or float(margin[flips].max()) < 2e-4 * scale  # indices agree wherever the decision is not a near-tie
    assert tuple(M.parsing_mask(sim_parsing, None, low, remap, (64, 64), (256, 256)).shape) == (1, 1, 256, 256)
    assert get_segmentation.__doc__ and callable(net.parse)


def test_glue_stencils_vs_reference_golden(simlib, golden):
    """BicubicDownSample (utils/bicubic.py) and DilateErosion.mask (utils/image_utils.py) as HIP kernels against golden
    vectors from the reference's own classes."""
    from hairfastgan_amd.hair_swap import BicubicDownSample

    G = golden("glue.npz")
    x = C.unit_input("glue/bicubic", (2, 3, 64, 64))
    for f in (2, 4):
        y = M.bicubic_down(simlib, None, x, BicubicDownSample(f).k, f)
        assert float((y - torch.from_numpy(G[f"bicubic{f}"])).abs().max()) < 2e-6
    mask = (C.unit_input("glue/mask", (3, 1, 48, 48)) > 0.3).float()
    d, e = M.dilate_erode(simlib, None, mask, 3)
    assert torch.equal(d, torch.from_numpy(G["dilate3"])) and torch.equal(e, torch.from_numpy(G["erode3"]))


def test_shape_adaptor_blocks_and_layout(simlib, monkeypatch, golden):
    """SURVEY section 8 row f4, the CtrlHair mask generator: (1) the oracle reproduces the reference's golden codes and
    label map of one pair bit for bit; (2) the HIP-backed mirror has the reference's state-dict layout; (3) its two
    non-standard blocks on the interpreted kernels - the 4x4 / stride-2 / pad-1 conv as a 3x3 conv of the space-to-depth
    input, and the per-sample LayerNorm (unbiased std, eps added to it) + LeakyReLU - against torch."""
    import sys

    from hairfastgan_amd import shape_adaptor as SAP
    from oracle import ref_shape_adaptor as SA

    G = golden("shape_adaptor.npz")
    P = C.shape_adaptor_params()
    m1, m2 = C.shape_masks()
    lab, logits, fc, hc = SA.adapt_shape(P, m1[:1], m2[:1])
    assert torch.equal(fc, torch.from_numpy(G["face_code_0"])) and torch.equal(hc, torch.from_numpy(G["hair_code_0"]))
    assert torch.equal(lab[0].to(torch.uint8), torch.from_numpy(G["labels"][0]))
    assert len(set(lab.flatten().tolist())) >= 4  # a map with several regions, not one class everywhere
    gen = SAP.MaskGenerator()
    sd = gen.state_dict()
    assert {k: tuple(v.shape) for k, v in sd.items()} == SA.param_shapes() and list(sd) == list(SA.param_shapes())
    assert torch.equal(gen.hair_encoder.input_embedding[0], SA.pos_embedding())

    for mod in (sys.modules["hairfastgan_amd.shape_adaptor"], sys.modules["hairfastgan_amd.encoders._fused"]):
        monkeypatch.setattr(mod, "lib", lambda: simlib)
        monkeypatch.setattr(mod, "stream", lambda: None)
        monkeypatch.setattr(mod, "require_gpu", lambda *a: None)
    torch.manual_seed(4)
    blk = SAP.Conv2dBlock(5, 24, 4, 2, padding=1, norm="ln", activation="lrelu")
    x = torch.randn(2, 5, 12, 20)
    ref = F.conv2d(F.pad(x, (1, 1, 1, 1)), blk.conv.weight, blk.conv.bias, stride=2)
    flat = ref.reshape(2, -1)
    ref = (ref - flat.mean(1).view(-1, 1, 1, 1)) / (flat.std(1).view(-1, 1, 1, 1) + 1e-5)
    ref = F.leaky_relu(ref * blk.norm.gamma.view(1, -1, 1, 1) + blk.norm.beta.view(1, -1, 1, 1), 0.2)
    ref = ref.detach()
    with torch.inference_mode():
        y = blk(x)
    assert y.shape == ref.shape and float((y - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    # the same layer with its last two input channels declared constant (a MaskEncoder's positional encoding): their
    # share of the conv is folded into a per-pixel bias at plan time
    blk._plan = None
    xc = torch.cat([x[:, :3], x[:1, 3:].expand(2, -1, -1, -1)], 1)
    with torch.no_grad():
        refc = F.conv2d(F.pad(xc, (1, 1, 1, 1)), blk.conv.weight, blk.conv.bias, stride=2)
        flat = refc.reshape(2, -1)
        refc = (refc - flat.mean(1).view(-1, 1, 1, 1)) / (flat.std(1).view(-1, 1, 1, 1) + 1e-5)
        refc = F.leaky_relu(refc * blk.norm.gamma.view(1, -1, 1, 1) + blk.norm.beta.view(1, -1, 1, 1), 0.2)
    with torch.inference_mode():
        yc = blk(x[:, :3].contiguous(), const_planes=x[:1, 3:].contiguous())
    assert float((yc - refc).abs().max()) < 2e-5 * max(1.0, float(refc.abs().max()))
    # deep-layer form: 8x8 input -> patches + GEMM on the 1x1 conv kernel
    blk8 = SAP.Conv2dBlock(6, 10, 4, 2, padding=1, norm="none", activation="none")
    x8 = torch.randn(3, 6, 8, 8)
    from hairfastgan_amd import _runtime
    with torch.no_grad():
        r8 = F.conv2d(F.pad(x8, (1, 1, 1, 1)), blk8.conv.weight, blk8.conv.bias, stride=2)
    with torch.inference_mode():  # batch-invariant plans (the default): the row-wise GEMV form at any batch (the GEMM folds the batch)
        y8 = blk8(x8)
    assert "gemm" not in blk8._plan and y8.shape == r8.shape and float((y8 - r8).abs().max()) < 2e-5
    prev = _runtime._batch_invariant
    _runtime._batch_invariant = False  # (host-side predicate only: the simulated library is called directly)
    try:
        with torch.inference_mode():
            y8 = blk8(x8)
    finally:
        _runtime._batch_invariant = prev
    assert "gemm" in blk8._plan and y8.shape == r8.shape and float((y8 - r8).abs().max()) < 2e-5
    with torch.inference_mode():  # few patch rows (a single swap): the weight-streaming linear kernel
        y8s = blk8(x8[:1, :, :4, :4].contiguous())
    with torch.no_grad():
        r8s = F.conv2d(F.pad(x8[:1, :, :4, :4], (1, 1, 1, 1)), blk8.conv.weight, blk8.conv.bias, stride=2)
    assert y8s.shape == r8s.shape and float((y8s - r8s).abs().max()) < 2e-5
    # a channel count that is padded up to the fp16 kernel's 64-channel tiles (16 input channels, 24 -> 64 filters)
    blk16 = SAP.Conv2dBlock(16, 24, 3, 1, padding=1, norm="ln", activation="lrelu")
    x16 = torch.randn(2, 16, 16, 32)
    with torch.no_grad():
        r16 = F.conv2d(x16, blk16.conv.weight, blk16.conv.bias, padding=1)
        flat = r16.reshape(2, -1)
        r16 = (r16 - flat.mean(1).view(-1, 1, 1, 1)) / (flat.std(1).view(-1, 1, 1, 1) + 1e-5)
        r16 = F.leaky_relu(r16 * blk16.norm.gamma.view(1, -1, 1, 1) + blk16.norm.beta.view(1, -1, 1, 1), 0.2)
    with torch.inference_mode():
        y16 = blk16(x16)
    assert blk16._plan["pad_to"] == 64 and y16.shape == r16.shape
    assert float((y16 - r16).abs().max()) < 2e-5 * max(1.0, float(r16.abs().max()))
    blk3 = SAP.Conv2dBlock(8, 3, 3, 1, padding=1, norm="none", activation="none")
    x3 = torch.randn(1, 8, 9, 11)
    with torch.inference_mode():
        y3 = blk3(x3)
    assert float((y3 - F.conv2d(x3, blk3.conv.weight, blk3.conv.bias, padding=1)).abs().max()) < 2e-5


@pytest.mark.parametrize("B,H,W", [(2, 64, 96), (1, 37, 51)])
def test_stem_conv_pool_on_fp16_cores(simlib, B, H, W):
    """hf_stem7x7s2_f16_f32 (csrc/stem.hip): conv 7x7/2 + affine + ReLU, alone and with the 3x3/2 max pool fused, against
    torch on planes with ragged tiles (odd sizes: the last tile row / column, the pooling pad at every border)."""
    torch.manual_seed(H)
    x = torch.randn(B, 3, H, W)
    w = torch.randn(64, 3, 7, 7) * 0.05
    sc, sh = torch.rand(64) + 0.5, torch.randn(64) * 0.3
    w3 = M.stem_prepare(w)
    ref = F.relu(F.conv2d(x, w, stride=2, padding=3) * sc[None, :, None, None] + sh[None, :, None, None])
    y = M.stem7x7s2(simlib, None, x, w3, out_scale=sc, bias=sh, alpha=0.0, pool=False)
    assert tuple(y.shape) == tuple(ref.shape) and float((y - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    yp = M.stem7x7s2(simlib, None, x, w3, out_scale=sc, bias=sh, alpha=0.0, pool=True)
    refp = F.max_pool2d(ref, 3, 2, 1)
    assert tuple(yp.shape) == tuple(refp.shape) and float((yp - refp).abs().max()) < 2e-5 * max(1.0, float(refp.abs().max()))
    assert torch.equal(yp, F.max_pool2d(y, 3, 2, 1))  # the fused pool = pooling the unfused result, bit for bit
    # leaky slope: negative values survive, the pooling pad must still behave like -inf
    yl = M.stem7x7s2(simlib, None, x, w3, out_scale=sc, bias=sh - 3.0, alpha=0.2, pool=True)
    refl = F.max_pool2d(F.leaky_relu(F.conv2d(x, w, stride=2, padding=3) * sc[None, :, None, None] + (sh - 3.0)[None, :, None, None], 0.2), 3, 2, 1)
    assert float((yl - refl).abs().max()) < 2e-5 * max(1.0, float(refl.abs().max()))
