"""GPU parity tests (-m gpu) of the face-parsing path (north_star: "bit-exact
segmentation-mask indices"): BiSeNet + get_segmentation on the HIP path against golden masks produced by the real
reference, and the mask of a HIP-generated 1024^2 image against the mask of the oracle-generated one.

A mask index is an argmax over 19 logits: it is reproduced exactly wherever the top-1 / top-2 margin exceeds the
fp32 tolerance of the logits (1e-4 x the largest logit).  The tests count the flips, require every one of them to sit
below that margin, and print the count."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cases as C
from oracle import ref_bisenet as BS
from oracle import ref_stylegan2 as O

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    return torch.device("cuda:0")


def _net(dev):
    from hairfastgan_amd.face_parsing import BiSeNet

    net = BiSeNet(19).eval()
    net.load_state_dict(C.bisenet_params())
    return net.to(dev)


@pytest.mark.parametrize("tag", ["512", "320x384"])
@pytest.mark.parametrize("mode", ["f16x3", "f32"])
def test_masks_vs_reference_golden(golden, tag, mode):
    from hairfastgan_amd import _runtime
    from hairfastgan_amd.face_parsing import get_segmentation

    dev = _dev()
    G = golden("bisenet.npz")
    net = _net(dev)
    x = C.bisenet_input(tag).to(dev)
    prev = _runtime.set_conv_precision(mode)
    try:
        with torch.inference_mode():
            low = net.logits_low(x)
            full = get_segmentation(net, x, resize=False)
            small = get_segmentation(net, x, resize=True)
            again = get_segmentation(net, x, resize=False)
    finally:
        _runtime.set_conv_precision(prev)
    assert torch.equal(full, again)
    H, W = x.shape[2:]
    logits = F.interpolate(low, (H, W), mode="bilinear", align_corners=True).cpu()
    f = logits.reshape(-1)
    step = max(1, f.numel() // 2048)
    ref_s = torch.from_numpy(G[f"logits_samples_{tag}"])
    scale = float(max(abs(G[f"logits_stats_{tag}"][2]), abs(G[f"logits_stats_{tag}"][3])))
    err = float((f[::step][:2048] - ref_s).abs().max())
    assert err <= 1e-4 * scale, (err, scale)
    margin = torch.from_numpy(G[f"margin_{tag}"].astype(np.float32))
    ref_full = torch.from_numpy(G[f"mask_{tag}"].astype(np.int64))
    flips = full[0, 0].cpu() != ref_full
    n = int(flips.sum())
    # north_star: "bit-exact segmentation-mask indices".  On the reference's golden inputs every index is reproduced
    # (observed 0 of 262 144 / 0 of 122 880 in both modes since round 2): assert exactly that, so that a regression
    # shows; the near-tie margin clause is kept only for the generated-image chain below.
    assert n == 0, (n, float(margin[flips].max()))
    ref_small = torch.from_numpy(G[f"mask256_{tag}"].astype(np.int64))
    n_small = int((small[0, 0].cpu() != ref_small).sum())
    assert n_small == 0
    assert tuple(small.shape) == (1, 1, 256, 256) and small.dtype == torch.int64
    print(f"bisenet {tag} {mode}: {n} of {H * W} full-resolution indices differ from the reference ({n_small} of 65536 in the "
          f"256^2 mask), all with top-1/top-2 margin <= {float(margin[flips].max()) if n else 0.0:.2e} (largest logit {scale:.1f}); "
          f"max logit error {err:.2e}")


def test_generated_image_mask_hip_vs_oracle():
    """End of the chain Alignment.py:63-67 uses: W+ -> 1024^2 image (generator, explicit noise) -> ((I+1)/2).clip(0,1) ->
    ImageNet normalisation -> get_segmentation.  HIP generator + HIP BiSeNet against oracle generator + oracle BiSeNet."""
    from hairfastgan_amd.face_parsing import get_segmentation
    from hairfastgan_amd.stylegan2.model import Generator

    dev = _dev()
    size, cm, n_mlp, _, _ = C.GENERATOR_CASES["g1024"]
    g = Generator(size, 512, n_mlp, channel_multiplier=cm).eval()
    shapes = {k: tuple(v.shape) for k, v in g.state_dict().items()}
    Pg = C.generator_params(shapes)
    g.load_state_dict(Pg)
    g = g.to(dev)
    lat, nz, _ = C.generator_inputs(size, 1, 0)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    net = _net(dev)
    with torch.inference_mode():
        img, _ = g([lat.to(dev)], input_is_latent=True, noise=[n.to(dev) for n in nz])
        x = (((img + 1) / 2).clip(0, 1) - mean.to(dev)) / std.to(dev)
        mask = get_segmentation(net, x)                     # [1,1,256,256]
        mask_full = get_segmentation(net, x, resize=False)  # [1,1,1024,1024]
    torch.set_num_threads(min(32, torch.get_num_threads()))
    img_o, _ = O.generator_forward(Pg, lat, nz)
    x_o = (((img_o + 1) / 2).clip(0, 1) - mean) / std
    Pb = C.bisenet_params()
    logits_o = BS.bisenet_logits(Pb, x_o)
    ref_full = torch.tensor(BS.LABEL_REMAP)[logits_o[0].argmax(0)]
    top2 = logits_o[0].topk(2, dim=0).values
    margin = top2[0] - top2[1]
    scale = float(logits_o.abs().max())
    flips = mask_full[0, 0].cpu() != ref_full
    n = int(flips.sum())
    assert float((img.cpu() - img_o).abs().max()) <= 1e-4 * max(1.0, float(img_o.abs().max()))
    assert n == 0 or float(margin[flips].max()) <= 5e-4 * scale, (n, float(margin[flips].max()), scale)
    ref_small = BS.get_segmentation(Pb, x_o, resize=True)
    n_small = int((mask.cpu() != ref_small).sum())
    assert n_small <= max(n, 0)
    print(f"generated 1024^2 image -> mask: {n} of {1024 * 1024} indices differ ({n_small} of 65536 in the 256^2 mask); "
          f"largest margin among them {float(margin[flips].max()) if n else 0.0:.2e} (largest logit {scale:.1f})")


def test_shape_adaptor_vs_reference_golden(golden):
    """the CtrlHair mask generator (get_hair_face_code + get_new_shape, Alignment.py:74-77) on the
    HIP kernels against the reference's golden codes and label maps: codes to 1e-4, label indices equal wherever the
    decision is not a near-tie of the two largest logits; both pairs in one batched call and one by one."""
    from hairfastgan_amd.shape_adaptor import MaskGenerator, adapt_shape, get_hair_face_code, get_new_shape

    dev = torch.device("cuda:0")
    G = golden("shape_adaptor.npz")
    gen = MaskGenerator().eval()
    gen.load_state_dict(C.shape_adaptor_params())
    gen.to(dev)
    m1, m2 = (m.to(dev) for m in C.shape_masks())
    out = adapt_shape(gen, m1, m2)
    assert out.shape == (2, 1, 256, 256) and out.dtype == torch.int64
    scale = float(G["logits_stats"][3]) if "logits_stats" in G.files else 1.0
    for b in range(2):
        face_code, _ = get_hair_face_code(gen, m1[b, 0])
        _, hair_code = get_hair_face_code(gen, m2[b, 0])
        for got, key in ((face_code, f"face_code_{b}"), (hair_code, f"hair_code_{b}")):
            ref = torch.from_numpy(G[key]).to(dev)
            assert float((got - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max())), key
        ref_lab = torch.from_numpy(G["labels"][b].astype("int64")).to(dev)
        margin = torch.from_numpy(G[f"margin_{b}"].astype("float32")).to(dev)
        for lab in (out[b, 0], get_new_shape(gen, face_code, hair_code)):
            flips = lab != ref_lab
            # (observed: 0 flips in f16x3 - the default -, 1 in f32 mode at a reference margin of 1.7e-5: tools/probes/shape_adaptor_flips.py,
            # profiles/r06q_shape_adaptor_flips.txt)
            assert int(flips.sum()) <= 2
            assert int(flips.sum()) == 0 or float(margin[flips].max()) < 1e-4 * max(1.0, abs(scale)), (int(flips.sum()), float(margin[flips].max()))


@pytest.mark.parametrize("B,H,W", [(2, 256, 256), (1, 203, 317)])
def test_stem_conv_pool_fp16_cores(B, H, W):
    """hf_stem7x7s2_f16_f32 (csrc/stem.hip) through the C ABI: conv1 + bn1 + ReLU + max pool of BiSeNet's ResNet stem in one
    pass, against torch's fp32 CPU convolution (tolerance 2e-5 of the largest value: f16x3 operands are fp32-class) and
    against the unfused result pooled by torch (bit-equal)."""
    from hairfastgan_amd import _marshal as M
    from hairfastgan_amd._runtime import lib, stream

    dev = _dev()
    torch.manual_seed(W)
    x = torch.randn(B, 3, H, W)
    w = torch.randn(64, 3, 7, 7) * 0.05
    sc, sh = torch.rand(64) + 0.5, torch.randn(64) * 0.3
    ref = F.relu(F.conv2d(x.double(), w.double(), stride=2, padding=3) * sc[None, :, None, None] + sh[None, :, None, None]).float()
    w3 = M.stem_prepare(w.to(dev))
    y = M.stem7x7s2(lib(), stream(), x.to(dev), w3, out_scale=sc.to(dev), bias=sh.to(dev), alpha=0.0, pool=False)
    yp = M.stem7x7s2(lib(), stream(), x.to(dev), w3, out_scale=sc.to(dev), bias=sh.to(dev), alpha=0.0, pool=True)
    tol = 2e-5 * max(1.0, float(ref.abs().max()))
    assert float((y.cpu() - ref).abs().max()) < tol
    assert float((yp.cpu() - F.max_pool2d(ref, 3, 2, 1)).abs().max()) < tol
    assert torch.equal(yp.cpu(), F.max_pool2d(y.cpu(), 3, 2, 1))
