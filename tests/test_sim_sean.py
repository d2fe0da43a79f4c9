"""CPU tests of the SEAN inpainting stage: the oracle against the golden vectors the real
reference produced, the product's state-dict layout, and - kernel sources interpreted by tests/hipsim - the whole
HIP-path mirror (Zencoder, table-lookup convolutions, ACE tail, SPADE blocks) on a scaled-down model against the oracle."""
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from hairfastgan_amd import _marshal as M
from oracle import cases as C
from oracle import ref_sean as SN
from oracle import synth


def test_oracle_reproduces_reference_codes(golden):
    """oracle/ref_sean.py encode_sean vs the golden style codes make_golden.py produced with the reference's
    Pix2PixModel / SPADEGenerator.Zencoder (the two decodes are pinned in oracle_vs_reference.json: 0.0; they take
    minutes on CPU and are compared on the GPU in tests/test_gpu_sean.py)."""
    G = golden("sean.npz")
    images, labels, _target, _noise = C.sean_inputs()
    codes = SN.encode_sean(C.sean_params(), images, labels)
    assert np.array_equal(codes.numpy(), G["codes"])
    absent = (codes == 0).all(-1)
    assert absent.any() and not absent.all()  # decode_sean's median-code rule is exercised by the fixture


def test_state_dict_layout_matches_reference():
    from hairfastgan_amd.sean import SeanModel

    with torch.device("meta"):
        m = SeanModel()
    mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    want = SN.sean_param_shapes()
    assert mine == want and list(mine) == list(want)


def test_label_conv_and_ace_tail(simlib):
    torch.manual_seed(0)
    labels = torch.randint(0, 19, (2, 9, 12), dtype=torch.int32)
    onehot = F.one_hot(labels.long(), 19).permute(0, 3, 1, 2).float()
    w, b = torch.randn(6, 19, 3, 3), torch.randn(6)
    table = w.permute(2, 3, 0, 1).reshape(54, 19).contiguous()
    y = M.label_conv3x3(simlib, None, labels, table, b, 6, relu=True)
    assert float((y - F.relu(F.conv2d(onehot, w, b, padding=1))).abs().max()) < 1e-5
    # per-sample region vectors, two samples per label map (group 2): conv of the broadcast vectors
    vec = torch.randn(4, 19, 5)                                    # [sample, label, feature]
    wg, bg = torch.randn(8, 5, 3, 3), torch.randn(8)
    tab = torch.einsum("ckyx,blk->yxcbl", wg, vec).reshape(72, 4 * 19).contiguous()
    ya = M.label_conv3x3(simlib, None, labels, tab, bg, 8, batch=4, cols_per_sample=19, group=2)
    for s in range(4):
        mid = vec[s][labels[s // 2].long()].permute(2, 0, 1)[None]  # [1,5,H,W]
        assert float((ya[s:s + 1] - F.conv2d(mid, wg, bg, padding=1)).abs().max()) < 1e-4
    # a plane large enough for the interior fast path (one lookup of the tap sum where a wave's 64 pixels all have a
    # single-label neighbourhood): blocks of one label with ragged borders -> both paths inside one launch
    big = torch.zeros(1, 32, 40, dtype=torch.int32)
    big[0, :, 20:] = 7
    big[0, 25:, :13] = 3
    big[0, 5, 5] = 11
    yb = M.label_conv3x3(simlib, None, big, table, b, 6, relu=False)
    ohb = F.one_hot(big.long(), 19).permute(0, 3, 1, 2).float()
    assert float((yb - F.conv2d(ohb, w, b, padding=1)).abs().max()) < 2e-5
    # labels outside 0..18 (a 255 'ignore' value, a negative one): both kernels clamp into the sample's own columns -
    # the per-pixel kernel (small plane) and the LDS-table kernel (>= 1024 pixels, W % 4 == 0) give the clamped map's result
    for shape in ((1, 9, 12), (1, 32, 40)):
        odd = torch.randint(0, 19, shape, dtype=torch.int32)
        odd[0, 2, 3], odd[0, 4, 4], odd[0, 0, 0] = 255, -3, 19
        want = M.label_conv3x3(simlib, None, odd.clamp(0, 18), table, b, 6)
        assert torch.equal(M.label_conv3x3(simlib, None, odd, table, b, 6), want)
    # ACE tail
    x, r = torch.randn(4, 4, 9, 12), torch.randn(4, 9, 12)
    nv, sc, sh = torch.randn(4) * 0.1, torch.rand(4) + 0.5, torch.randn(4)
    sp, avg, blend = torch.randn(2, 8, 9, 12), torch.randn(4, 8, 9, 12), torch.tensor([0.3, -0.7])
    n = (x + r[:, None] * nv[None, :, None, None]) * sc[None, :, None, None] + sh[None, :, None, None]
    spx = sp.repeat_interleave(2, 0)
    ag, ab = torch.sigmoid(blend[0]), torch.sigmoid(blend[1])
    want = n * (1 + ag * avg[:, :4] + (1 - ag) * spx[:, :4]) + ab * avg[:, 4:] + (1 - ab) * spx[:, 4:]
    got = M.ace_modulate(simlib, None, x, r, nv, sc, sh, avg, sp, blend, group=2, slope=0.2)
    assert float((got - F.leaky_relu(want, 0.2)).abs().max()) < 1e-5
    got = M.ace_modulate(simlib, None, x, None, None, sc, sh, None, sp, None, group=2, slope=1.0)
    want = (x * sc[None, :, None, None] + sh[None, :, None, None]) * (1 + spx[:, :4]) + spx[:, 4:]
    assert float((got - want).abs().max()) < 1e-5
    # the avg planes looked up INSIDE the tail kernel (hf_ace_modulate_table_f32) = label conv then tail, bit for bit;
    # plane with interiors, borders and a ragged last block (32 x 40 = 1280 pixels: the four-pixel LDS-table kernels)
    lab2 = torch.stack([big[0], torch.randint(0, 19, (32, 40), dtype=torch.int32)])
    lab2[1, 8:24, 8:32] = 2
    xb, rb = torch.randn(4, 4, 32, 40), torch.randn(4, 32, 40)
    spb = torch.randn(2, 8, 32, 40)
    avg_b = M.label_conv3x3(simlib, None, lab2, tab, bg, 8, batch=4, cols_per_sample=19, group=2)
    for s_ in range(4):
        mid = vec[s_][lab2[s_ // 2].long()].permute(2, 0, 1)[None]
        assert float((avg_b[s_:s_ + 1] - F.conv2d(mid, wg, bg, padding=1)).abs().max()) < 1e-4
    two = M.ace_modulate(simlib, None, xb, rb, nv, sc, sh, avg_b, spb, blend, group=2, slope=0.2)
    one = M.ace_modulate_table(simlib, None, xb, rb, nv, sc, sh, lab2, tab, bg, spb, blend, group=2, slope=0.2)
    assert torch.equal(one, two)
    # x_up: the tails read a half-resolution x in place = the materialised nearest x2 up-sampling, bit for bit (both kernels)
    x_lo = torch.randn(4, 4, 16, 20)
    x_hi = x_lo.repeat_interleave(2, 2).repeat_interleave(2, 3).contiguous()
    assert torch.equal(M.ace_modulate_table(simlib, None, x_lo, rb, nv, sc, sh, lab2, tab, bg, spb, blend, group=2, slope=0.2, x_up=True),
                       M.ace_modulate_table(simlib, None, x_hi, rb, nv, sc, sh, lab2, tab, bg, spb, blend, group=2, slope=0.2))
    assert torch.equal(M.ace_modulate(simlib, None, x_lo, rb, nv, sc, sh, None, spb, None, group=2, slope=1.0, x_up=True),
                       M.ace_modulate(simlib, None, x_hi, rb, nv, sc, sh, None, spb, None, group=2, slope=1.0))
    # region pooling with tanh, on the interior of a padded plane
    xp = torch.randn(2, 3, 11, 14)
    lab = labels.clone()
    lab[1][lab[1] == 4] = 5  # label 4 absent from sample 1
    got = M.region_mean(simlib, None, xp, lab, crop=1, act_tanh=True)
    t = torch.tanh(xp[:, :, 1:-1, 1:-1])
    for s in range(2):
        for l in range(19):
            m = lab[s] == l
            want = t[s][:, m].mean(1) if m.any() else torch.zeros(3)
            assert float((got[s, l] - want).abs().max()) < 1e-6
    assert torch.equal(M.tanh(simlib, None, xp), torch.tanh(xp)) or float((M.tanh(simlib, None, xp) - torch.tanh(xp)).abs().max()) < 1e-6


@pytest.fixture()
def sim_sean(simlib, monkeypatch):
    import hairfastgan_amd.encoders  # noqa: F401
    import hairfastgan_amd.sean  # noqa: F401

    for n in ("hairfastgan_amd.encoders._fused", "hairfastgan_amd.sean"):
        mod = sys.modules[n]
        monkeypatch.setattr(mod, "lib", lambda: simlib)
        monkeypatch.setattr(mod, "stream", lambda: None)
        monkeypatch.setattr(mod, "require_gpu", lambda *a: None)
    return simlib


def test_small_model_vs_oracle(sim_sean):
    """A scaled-down SEAN (ngf 1, style length 8, 64^2) through the product's code path - encode of a pair, both decodes
    as one batch sharing the target mask, explicit ACE noise - against the oracle on the same parameters."""
    from hairfastgan_amd.sean import SeanModel

    cfg = SN.Cfg(ngf=1, style=8, hidden=4, size=64, zc=(4, 8, 8, 8))
    P = C.sean_params(cfg)
    mean_codes = C.t(synth.pseudo_normal("sean/small/mean", (19, 8))) * 0.5
    m = SeanModel(mean_codes, ngf=1, style=8, hidden=4, size=64, zencoder_widths=(4, 8, 8, 8)).eval()
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == SN.sean_param_shapes(cfg=cfg)
    m.load_state_dict(P)
    m1, m2 = C.shape_masks()
    down = lambda t: t[:, :, ::4, ::4].contiguous()  # noqa: E731
    labels = torch.cat([down(m1[0:1]), down(m2[0:1])])
    labels[1][labels[1] == 2] = 1
    target = down(m1[1:2])
    images = C.t(synth.uniform01("sean/small/images", (2, 3, 64, 64)))
    codes = m.encode(images, labels)
    codes_o = SN.encode_sean(P, images, labels)
    assert float((codes - codes_o).abs().max()) < 2e-5
    assert (codes_o == 0).all(-1).any()
    order = SN.ace_call_order(cfg)
    noise = [C.t(synth.pseudo_normal(f"sean/small/noise/{i}", (2, r, r, 1))) for i, (_b, _a, _c, r) in enumerate(order)]
    taps, taps_o = {}, {}
    got = m.decode(codes_o, target, group=2, noise=[n[..., 0].transpose(1, 2).contiguous() for n in noise], taps=taps)
    want = SN.spade_generator(P, SN.one_hot(target).expand(2, -1, -1, -1), SN.merge_codes(codes_o, mean_codes), noise, taps=taps_o, cfg=cfg)
    for name in taps_o:
        ref = taps_o[name] if name != "up_3" else F.leaky_relu(taps_o[name], 0.2)  # the last block emits its LeakyReLU
        assert float((taps[name] - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max())), name
    assert got.shape == (2, 3, 64, 64) and float((got - want).abs().max()) < 2e-5
    assert float((got[0] - got[1]).abs().max()) > 1e-3  # the two decodes differ through their style codes only
