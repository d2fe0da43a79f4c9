"""GPU test (-m gpu) of the CLIP ViT-B/32 image tower at its real size against the oracle
restatement (parity unpinned against the reference: the `clip` package is an un-vendored dependency - oracle/ref_clip.py),
and of ClipBlendingModel with the native tower inside against the oracle composition."""
import pytest
import torch

from oracle import cases as C
from oracle import ref_clip as RC
from oracle import ref_postprocess as PP
from oracle import synth

pytestmark = pytest.mark.gpu


def test_clip_tower_vs_oracle():
    from hairfastgan_amd.clip_vit import ClipImageTower

    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    dev = torch.device("cuda:0")
    P = C.clip_params()
    m = ClipImageTower().eval()
    m.load_clip_state_dict(P)
    m.to(dev)
    img = C.t(synth.pseudo_normal("clip/img", (3, 3, 224, 224)))
    taps, taps_o = {}, {}
    got = m.visual(img.to(dev), taps=taps)
    want = RC.encode_image(P, img, taps=taps_o)
    for i in (0, 5, 11):
        ref = taps_o[i]
        err = float((taps[i][0].permute(2, 1, 0).cpu() - ref).abs().max())
        assert err < 1e-4 * max(1.0, float(ref.abs().max())), (i, err)
    err = float((got.cpu() - want).abs().max())
    assert got.shape == (3, 512) and err < 1e-4 * max(1.0, float(want.abs().max())), err
    # batch independence: row 1 alone
    alone = m.encode_image(img[1:2].to(dev))
    assert float((alone[0] - got[1]).abs().max()) < 1e-5
    # ClipBlendingModel around the native tower == the oracle composition around the oracle tower
    from hairfastgan_amd.encoders import ClipBlendingModel

    Pb = C.params_from_shapes("clipblend", PP.clip_blending_param_shapes())
    blend = ClipBlendingModel(image_embed=m.encode_image).eval()
    blend.load_state_dict(Pb)
    blend.to(dev)
    _, _, s_face, s_color, img_face, img_color = C.latent_model_inputs()
    out = blend(s_face.to(dev), s_color.to(dev), img_face.to(dev), img_color.to(dev))
    ref = PP.clip_blending(Pb, s_face, s_color, img_face, img_color, lambda x: RC.encode_image(P, x))
    assert float((out.cpu() - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))
    print(f"clip tower: max-abs vs oracle {err:.2e}")
