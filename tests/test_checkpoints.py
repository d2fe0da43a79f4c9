"""`HairFast(args)` as a drop-in (round-3 verdict item 1): with no state dicts passed, every network is read from the
reference's checkpoint files - the paths of hair_swap.py:108-133's parser and the ones the reference hard-codes - with the
reference's key handling, and a missing file raises FileNotFoundError naming it: never random weights.  CPU only (modules are
constructed and loaded, no forward runs); tests/test_gpu_checkpoints.py runs a swap from such a tree."""
import os

import numpy as np
import pytest
import torch

from hairfastgan_amd import checkpoints as CK
from hairfastgan_amd.hair_swap import HairFast, get_parser
from tests import ckpt_tree as T


@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("reference_tree"))
    st = T.cheap_states()
    T.write_reference_tree(root, st, clip_mode="in_checkpoint")
    return root, st


def _same(module_state, want, skip=()):
    keys = [k for k in module_state if not any(k.startswith(s_) for s_ in skip)]
    assert sorted(keys) == sorted(want), (sorted(set(keys) ^ set(want))[:6])
    for k in keys:
        assert torch.equal(module_state[k].cpu().float(), want[k].float()), k


def test_hairfast_reads_the_reference_tree(tree):
    root, st = tree
    args = get_parser().parse_args([])
    args.device = "cpu"
    hf = HairFast(args, pretrained_root=root)
    _same(hf.net.generator.state_dict(), st["generator"])
    assert torch.equal(hf.net.latent_avg, st["generator_latent_avg"])
    _same(hf.embed.e4e.encoder.state_dict(), st["e4e"])                       # 'encoder.' prefix stripped, decoder entries dropped
    assert torch.equal(hf.embed.e4e.latent_avg, st["e4e_latent_avg"])
    _same(hf.embed.encoder.enc.state_dict(), st["fs"])
    assert torch.equal(hf.embed.encoder.dlatent_avg, st["fs_dlatent_avg"])   # psp_ffhq_encode.pt['latent_avg']
    _same(hf.parsing.state_dict(), st["bisenet"])
    assert hf.embed.parsing is hf.parsing and hf.align.parsing is hf.parsing  # the reference's singleton
    S = hf.stages
    _same(S.sean_model.netG.state_dict(), st["sean"])
    assert torch.equal(S.sean_model.mean_codes, st["sean_mean_codes"])        # the nineteen ACE.npy files
    _same(S.mask_generator.state_dict(), st["shape"])
    _same(S.rotate_model.state_dict(), st["rotate"])                          # ['model_state_dict']
    _same(S.blend_model.state_dict(), st["blend"])                            # clip_model.* dropped
    _same(S.clip_tower.state_dict(), st["clip"])                              # ... and loaded into the native tower
    _same(hf.blend.post_process.state_dict(), st["pp"])
    assert torch.equal(hf.blend.post_process.latent_avg, st["pp_latent_avg"].reshape(18, 512))
    for p_ in hf.net.generator.parameters():
        assert not p_.requires_grad                                            # models/Net.py:44-46


def test_args_paths_are_honoured(tree, tmp_path, monkeypatch):
    """--ckpt / --rotate_checkpoint / --blending_checkpoint / --pp_checkpoint are read from where the parser says, by the
    stage object that owns them in the reference (Alignment.py:29-38: SEAN, shape adaptor, Rotate; Blending.py:24-30)."""
    root, st = tree
    moved = {}
    for arg, rel in (("rotate_checkpoint", "pretrained_models/Rotate/rotate_best.pth"),
                     ("pp_checkpoint", "pretrained_models/PostProcess/pp_model.pth"),
                     ("blending_checkpoint", "pretrained_models/Blending/checkpoint.pth"),
                     ("ckpt", "pretrained_models/StyleGAN/ffhq.pt")):
        moved[arg] = str(tmp_path / (arg + ".pth"))
        os.symlink(os.path.join(root, rel), moved[arg])
    args = get_parser().parse_args([x for arg, path in moved.items() for x in ("--" + arg, path)])
    args.device = "meta"  # which files are opened is the subject: nothing is materialised
    seen = []
    real = CK.load_file
    monkeypatch.setattr(CK, "load_file", lambda path, what, root=None: (seen.append(str(path)), real(path, what, root))[1])
    from hairfastgan_amd import net as N
    from hairfastgan_amd.hair_swap import Alignment, Blending

    monkeypatch.setattr(N, "load_file", CK.load_file)
    net = N.Net(args, root=root)
    assert seen == [moved["ckpt"]]
    del seen[:]
    al = Alignment(args, net=net, parsing=object(), pretrained_root=root)
    assert seen == [CK.SEAN_PATH, CK.SHAPE_ADAPTOR_PATH, moved["rotate_checkpoint"]]
    assert al.stages.blend_model is None and al.stages.sean_model is not None and al.stages.rotate_model is not None
    del seen[:]
    bl = Blending(args, net=net, pretrained_root=root)
    assert seen == [moved["blending_checkpoint"], moved["pp_checkpoint"], CK.PP_LATENT_AVG_PATH]
    assert bl.stages.blend_model is not None and bl.stages.rotate_model is None and bl.stages.clip_tower is not None


@pytest.mark.parametrize("rel,what", [
    ("pretrained_models/StyleGAN/ffhq.pt", "StyleGAN2"),
    (CK.E4E_PATH, "e4e"), (CK.FS_ENCODER_PATH, "FeatureStyle"), (CK.FS_STYLEGAN_PATH, "dlatent_avg"),
    (CK.BISENET_PATH, "BiSeNet"), (CK.SEAN_PATH, "SEAN"), (CK.SEAN_CODES_DIR + "/13/ACE.npy", "label 13"),
    (CK.SHAPE_ADAPTOR_PATH, "shape adaptor"), ("pretrained_models/Rotate/rotate_best.pth", "RotateModel"),
    ("pretrained_models/Blending/checkpoint.pth", "ClipBlendingModel"),
    ("pretrained_models/PostProcess/pp_model.pth", "PostProcessModel"), (CK.PP_LATENT_AVG_PATH, "latent_avg"),
])
def test_a_missing_file_raises_and_names_it(tree, rel, what):
    root, _ = tree
    path = os.path.join(root, rel)
    os.rename(path, path + ".away")
    try:
        args = get_parser().parse_args([])
        args.device = "meta"  # nothing is materialised: the constructor must fail on the file, not on a device
        with pytest.raises(FileNotFoundError) as err:
            HairFast(args, pretrained_root=root)
        assert os.path.basename(rel) in str(err.value) and what in str(err.value)
    finally:
        os.rename(path + ".away", path)


def test_clip_tower_sources(tree, tmp_path, monkeypatch):
    """Blending.py:24-26: the tower is clip.load(...)'s model overwritten by the checkpoint's clip_model.* entries when it
    has them; otherwise clip.load's cached TorchScript archive; otherwise an error - never a random tower."""
    root, st = tree
    small = {k: v for k, v in list(st["clip"].items())[:6]}
    own, tower = CK.blending("pretrained_models/Blending/checkpoint.pth", root)
    assert set(own) == set(st["blend"]) and set(tower) == {k for k in st["clip"]}
    # a checkpoint without the tower + the TorchScript archive clip.load caches
    ck = {"model_state_dict": dict(st["blend"])}
    torch.save(ck, tmp_path / "blend_only.pth")
    monkeypatch.setenv("HAIRFAST_CLIP_WEIGHTS", str(tmp_path / "nowhere" / "ViT-B-32.pt"))
    with pytest.raises(FileNotFoundError) as err:
        CK.blending(str(tmp_path / "blend_only.pth"), root)
    assert "ViT-B-32.pt" in str(err.value)
    archive = tmp_path / "ViT-B-32.pt"
    torch.jit.script(T._module_tree({**small, "logit_scale": torch.tensor(4.6)})).save(str(archive))
    monkeypatch.setenv("HAIRFAST_CLIP_WEIGHTS", str(archive))
    own, tower = CK.blending(str(tmp_path / "blend_only.pth"), root)
    assert all(torch.equal(tower[k], v) for k, v in small.items())
    torch.save(small, archive)  # a plain state dict under the same name
    own, tower = CK.blending(str(tmp_path / "blend_only.pth"), root)
    assert all(torch.equal(tower[k], v) for k, v in small.items())
    with pytest.raises(NotImplementedError):
        CK.clip_tower("RN50")


def test_state_dicts_without_their_companions_raise():
    """No silent zero defaults: e4e / FS / PostProcess average latents and SEAN's median codes (round-3 advisor, medium)."""
    from hairfastgan_amd import hair_swap as H

    args = get_parser().parse_args([])
    args.device = "meta"
    with pytest.raises(ValueError, match="e4e_latent_avg"):
        H.build_e4e(args, state={})
    with pytest.raises(ValueError, match="fs_dlatent_avg"):
        H.build_fs_encoder(args, None, state={})
    with pytest.raises(ValueError, match="sean_mean_codes"):
        H.NativeLatentStages(H.Stages(), "meta", sean_state={})
    with pytest.raises(ValueError, match=r"\[19, 512\]"):
        from hairfastgan_amd.sean import SeanModel
        SeanModel(torch.zeros(18, 512))
    with pytest.raises(ValueError, match="clip_state"):
        H.NativeLatentStages(H.Stages(), "meta", blend_state={})
    with pytest.raises(ValueError, match="pp_latent_avg"):
        H.Blending(args, net=object(), stages=H.Stages(), pp_state={})


def test_loader_refuses_wrong_wrappers(tmp_path):
    torch.save({"weights": {}}, tmp_path / "r.pth")
    with pytest.raises(KeyError, match="model_state_dict"):
        CK.rotate(str(tmp_path / "r.pth"))
    torch.save({"state_dict": {"decoder.x": torch.zeros(1)}, "latent_avg": torch.zeros(18, 512)}, tmp_path / "e.pt")
    with pytest.raises(KeyError, match="encoder"):
        CK.e4e(path=str(tmp_path / "e.pt"))
    np.save(tmp_path / "x.npy", np.zeros(3))


class _NotAWeight:  # a global that torch.load(weights_only=True) refuses
    def __init__(self):
        self.x = 1


def test_full_unpickling_is_opt_in(tmp_path, monkeypatch):
    """Round-4 advisor: a checkpoint that fails safe loading is NOT silently unpickled (arbitrary code execution): the
    loader raises, naming the file and the opt-in; HAIRFAST_UNSAFE_LOAD=1 restores the reference's plain torch.load."""
    import pickle

    p = tmp_path / "odd.pth"
    torch.save({"model_state_dict": {"w": torch.ones(2)}, "extra": _NotAWeight()}, p)
    monkeypatch.delenv("HAIRFAST_UNSAFE_LOAD", raising=False)
    with pytest.raises(pickle.UnpicklingError, match="HAIRFAST_UNSAFE_LOAD=1") as err:
        CK.load_file(str(p), "RotateModel")
    assert "odd.pth" in str(err.value)
    monkeypatch.setenv("HAIRFAST_UNSAFE_LOAD", "1")
    with pytest.warns(UserWarning, match="full unpickling"):
        got = CK.load_file(str(p), "RotateModel")
    assert torch.equal(got["model_state_dict"]["w"], torch.ones(2)) and isinstance(got["extra"], _NotAWeight)


def test_modulation_stack_plan_follows_in_place_updates():
    """Round-4 advisor: the stacked copies of the first Linears / LayerNorms are re-made when a source parameter is
    updated in place, re-typed or re-loaded through a sub-module (key = storage, dtype, version), and the
    load_state_dict hook is a module-level function (a lambda would make torch.save(module) fail)."""
    import io
    import pickle

    from hairfastgan_amd.encoders import post_process as PPM

    m = PPM.ModulationModule(6)
    pickle.dumps(m)  # picklable with its hook
    torch.save(m, io.BytesIO())
    p = m.gamma_function[0].weight
    k0 = PPM._param_version(p)
    with torch.no_grad():
        p.mul_(2.0)  # what an optimizer step / param.copy_ does (writes through `.data` bypass every version counter)
    assert PPM._param_version(p) != k0
    k1 = PPM._param_version(p)
    m.gamma_function[0].load_state_dict(m.gamma_function[0].state_dict())  # a sub-module load: no ModulationModule hook fires
    assert PPM._param_version(p) != k1
    gen = PPM._STACK_GENERATION[0]
    m.load_state_dict(m.state_dict())
    assert PPM._STACK_GENERATION[0] == gen + 1
    with torch.inference_mode():
        t = torch.ones(3)
    assert PPM._param_version(t)[2] is None  # inference tensors carry no version counter
