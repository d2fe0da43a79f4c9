"""`HairFast(args)` reading the reference's checkpoint files gives the SAME swap, bit for bit, as the constructor that is
handed the state dicts (round-3 verdict item 1): the closed-form synthetic parameters are written to a temp tree in the
reference's on-disk layout (tests/ckpt_tree.py: wrappers, `encoder.` prefix, decoy entries, ACE.npy codes, the CLIP tower
inside the blending checkpoint) and loaded back by the file path alone."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _states():
    from oracle import cases as C
    from oracle import ref_encoders as E
    from oracle import ref_postprocess as PP
    from oracle import ref_stylegan2 as O

    pp_shapes = dict(PP.post_process_param_shapes())
    lat_shape = pp_shapes.pop("latent_avg")
    _, e4e_avg = C.e4e_inputs(2)
    _, dlat = C.fs_inputs(2)
    return {
        "generator": C.generator_params(O.generator_param_shapes(1024, 512, 8, 2)), "generator_latent_avg": torch.zeros(512),
        "e4e": C.params_from_shapes("e4e", E.e4e_param_shapes()), "e4e_latent_avg": e4e_avg,
        "fs": C.params_from_shapes("fs", E.fs_param_shapes()), "fs_dlatent_avg": dlat,
        "pp": C.params_from_shapes("pp", pp_shapes),
        "pp_latent_avg": (C.params_from_shapes("pp", {"latent_avg": lat_shape})["latent_avg"] * 0.1).reshape(1, 18, 512),
        "bisenet": C.pipeline_bisenet_params(), "rotate": C.params_from_shapes("rotate", PP.rotate_param_shapes()),
        "blend": C.params_from_shapes("clipblend", PP.clip_blending_param_shapes()), "clip": C.clip_params(),
        "shape": C.shape_adaptor_params(), "sean": C.sean_params(), "sean_mean_codes": C.sean_mean_codes(),
    }


@pytest.mark.parametrize("clip_mode", ["in_checkpoint", "jit_cache"])
def test_swap_from_checkpoint_files_equals_swap_from_state_dicts(tmp_path, monkeypatch, clip_mode):
    from hairfastgan_amd.hair_swap import HairFast, get_parser
    from tests import ckpt_tree as T

    st = _states()
    clip_path = T.write_reference_tree(str(tmp_path), st, clip_mode=clip_mode)
    if clip_path:
        monkeypatch.setenv("HAIRFAST_CLIP_WEIGHTS", clip_path)
    args = get_parser().parse_args([])
    args.device = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    face, shape, color = (torch.randint(0, 256, (3, 1024, 1024), dtype=torch.uint8, generator=g) for _ in range(3))

    hf_files = HairFast(args, pretrained_root=str(tmp_path))  # no state dict passed: the reference's constructor
    out_files = hf_files.swap(face, shape, color, seed=11).clone()
    del hf_files
    torch.cuda.empty_cache()
    hf_mem = HairFast(args, generator_state={"g_ema": st["generator"], "latent_avg": st["generator_latent_avg"]},
                      e4e_state=st["e4e"], e4e_latent_avg=st["e4e_latent_avg"], fs_state=st["fs"], fs_dlatent_avg=st["fs_dlatent_avg"],
                      pp_state=st["pp"], pp_latent_avg=st["pp_latent_avg"], bisenet_state=st["bisenet"], rotate_state=st["rotate"],
                      blend_state=st["blend"], clip_state=st["clip"], shape_state=st["shape"], sean_state=st["sean"],
                      sean_mean_codes=st["sean_mean_codes"])
    out_mem = hf_mem.swap(face, shape, color, seed=11)
    assert out_files.shape == (3, 1024, 1024) and torch.isfinite(out_files).all()
    assert torch.equal(out_files, out_mem)
    assert float(out_files.std()) > 1e-3  # not a constant image


def test_check_checkpoint_tool_runs_a_whole_swap_from_files(tmp_path, capsys):
    """tools/check_checkpoint.py --swap: the real-checkpoint validation a user runs first (f32 / f16x3 / f16 swaps from the
    reference's files through the per-object precision switch; clamp counter, image and mask agreement) - exercised on the
    synthetic tree: nothing clamps, the f16x3 image is fp32-class, the tool reports success."""
    import argparse
    import importlib.util
    import os

    from tests import ckpt_tree as T

    T.write_reference_tree(str(tmp_path), _states(), clip_mode="in_checkpoint")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("check_checkpoint", os.path.join(root, "tools", "check_checkpoint.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    rc = tool.check_swap(argparse.Namespace(pretrained_root=str(tmp_path), images=None, seed=3))
    text = capsys.readouterr().out
    assert rc == 0 and "f16x3 is safe on this swap" in text, text
    line = [ln for ln in text.splitlines() if ln.startswith("f16x3")][0]
    assert "clamped elements 0" in line
