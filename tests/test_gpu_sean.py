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
