"""GPU parity tests (-m gpu) of the SEAN inpainting stage against golden vectors produced
by the reference's own Pix2PixModel / SPADEGenerator (oracle/make_golden.py --only sean)."""
import numpy as np
import pytest
import torch

from oracle import cases as C
from oracle import ref_sean as SN

pytestmark = pytest.mark.gpu


def _model(dev):
    from hairfastgan_amd.sean import SeanModel

    m = SeanModel(C.sean_mean_codes()).eval()
    m.load_state_dict(C.sean_params())
    return m.to(dev)


def _close(got, ref, tol=1e-4, what=""):
    got, ref = got.detach().cpu().double(), torch.as_tensor(ref).double()
    err = float((got - ref).abs().max())
    scale = max(1.0, float(ref.abs().max()))
    assert err <= tol * scale, (what, err, scale)
    rms = float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt().clamp_min(1e-12))
    assert rms <= 3e-5, (what, rms)
    return err


@pytest.mark.parametrize("mode", ["f16x3", "f32"])
def test_sean_vs_reference_golden(golden, mode):
    from hairfastgan_amd import _runtime

    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    dev = torch.device("cuda:0")
    G = golden("sean.npz")
    m = _model(dev)
    images, labels, target, noise = C.sean_inputs()
    prev = _runtime.set_conv_precision(mode)
    try:
        codes = m.encode(images.to(dev), labels.to(dev))
        e_codes = _close(codes, G["codes"], what="codes")
        assert torch.equal((codes == 0).all(-1).cpu(), torch.from_numpy((G["codes"] == 0).all(-1)))
        ref_codes = torch.from_numpy(G["codes"]).to(dev)
        nat = lambda ns: [n[..., 0].transpose(1, 2).contiguous().to(dev) for n in ns]  # noqa: E731  [1,W,H,1] -> [1,H,W]
        outs = []
        for d in range(2):  # one decode at a time, as decode_sean is called (Alignment.py:130-131)
            taps = {}
            img = m.decode(ref_codes[d:d + 1], target.to(dev), noise=nat(noise[d]), taps=taps)
            outs.append(img)
            f = img[0].reshape(-1)
            step = max(1, f.numel() // 2048)
            _close(f[::step][:2048], G[f"gen{d}_samples"], what=f"gen{d} samples")
            _close(img[0, :, 96:160, 96:160], G[f"gen{d}_crop"], what=f"gen{d} crop")
            _close(img[0, :, :32, :32], G[f"gen{d}_corner"], what=f"gen{d} corner")
            st = G[f"gen{d}_stats"]
            assert abs(float(img.mean()) - st[0]) < 1e-5 and abs(float(img.std()) - st[1]) < 1e-5
            for name, x in taps.items():
                ref = G[f"gen{d}_tap_{name}"]
                if name == "up_3":
                    continue  # emitted with the final LeakyReLU applied (rides in conv_1's epilogue); the image checks cover it
                fx = x.reshape(-1)
                stp = max(1, fx.numel() // 64)
                _close(fx[::stp][:64], ref[4:], what=f"tap {name}")
        # both decodes of the pair as ONE batch sharing the target mask (what HairFast runs): same results
        noise2 = [torch.cat([a, b]) for a, b in zip(nat(noise[0]), nat(noise[1]))]
        both = m.decode(ref_codes, target.to(dev), group=2, noise=noise2)
        for d in range(2):
            assert float((both[d] - outs[d][0]).abs().max()) < 2e-5
        # the whole stage from images (fresh noise): shapes, range, determinism under a seed
        torch.manual_seed(3)
        a = m.inpaint_pairs(images.to(dev), labels.to(dev), target.to(dev))
        torch.manual_seed(3)
        b = m.inpaint_pairs(images.to(dev), labels.to(dev), target.to(dev))
        assert a.shape == (2, 3, 256, 256) and torch.equal(a, b) and float(a.abs().max()) <= 1.0
    finally:
        _runtime.set_conv_precision(prev)
    print(f"sean {mode}: codes max-abs {e_codes:.2e}")


def test_ace_tail_with_table_lookup_equals_two_kernel_form():
    """hf_ace_modulate_table_f32 (avg planes looked up in-kernel) = hf_label_conv3x3_f32 + hf_ace_modulate_f32 bit for bit
    at a decode's own size (8 samples, 256 channels, 128^2, two samples per label map), and the LDS-table label conv against
    torch's convolution of the broadcast region vectors."""
    import torch.nn.functional as F

    from hairfastgan_amd import _marshal as M
    from hairfastgan_amd._runtime import lib, stream

    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    B, Cc, H, W, group = 8, 256, 128, 128, 2
    labels = torch.randint(0, 19, (B // group, H // 8, W // 8), device=dev).repeat_interleave(8, 1).repeat_interleave(8, 2)
    labels = labels.to(torch.int32).contiguous()                       # 8 x 8 blocks: interiors and borders
    table = torch.randn(9 * 2 * Cc, B * 19, device=dev)
    b_avg = torch.randn(2 * Cc, device=dev)
    x, noise = torch.randn(B, Cc, H, W, device=dev), torch.randn(B, H, W, device=dev)
    sp = torch.randn(B // group, 2 * Cc, H, W, device=dev)
    nv, sc, sh = torch.randn(Cc, device=dev) * 0.1, torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev)
    blend = torch.tensor([0.3, -0.7], device=dev)
    L, st = lib(), stream()
    avg = M.label_conv3x3(L, st, labels, table, b_avg, 2 * Cc, batch=B, cols_per_sample=19, group=group)
    two = M.ace_modulate(L, st, x, noise, nv, sc, sh, avg, sp, blend, group=group, slope=0.2)
    one = M.ace_modulate_table(L, st, x, noise, nv, sc, sh, labels, table, b_avg, sp, blend, group=group, slope=0.2)
    torch.cuda.synchronize()
    assert torch.equal(one, two)
    # the lookup conv itself: sample 5's table columns as a dense conv of its one-hot map
    s_ = 5
    onehot = F.one_hot(labels[s_ // group].long(), 19).permute(2, 0, 1).float()[None]
    wdense = table[:, s_ * 19:(s_ + 1) * 19].reshape(3, 3, 2 * Cc, 19).permute(2, 3, 0, 1).contiguous()
    ref = F.conv2d(onehot.cpu().double(), wdense.cpu().double(), b_avg.cpu().double(), padding=1).float()
    assert float((avg[s_:s_ + 1].cpu() - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))
