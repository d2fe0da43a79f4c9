"""GPU test (-m gpu) of the Net boundary and the hot-path call schedule of one swap
(BASELINE.json configs[2]): shapes, return conventions and agreement of the composed
calls with the individually verified pieces."""
import argparse

import pytest
import torch

from oracle import cases as C
from oracle import ref_encoders as E
from oracle import ref_stylegan2 as O

pytestmark = pytest.mark.gpu


def test_swap_schedule_shapes_and_consistency():
    from hairfastgan_amd.hair_swap import HairFastHotPath, get_parser

    assert torch.cuda.is_available()
    args = get_parser().parse_args([])
    assert (args.size, args.latent, args.n_mlp, args.channel_multiplier, args.batch_size) == (1024, 512, 8, 2, 3)
    dev = torch.device("cuda:0")
    args.device = dev
    gen_shapes = O.generator_param_shapes(1024, 512, 8, 2)
    state = {"g_ema": C.generator_params(gen_shapes), "latent_avg": torch.zeros(512)}
    x256, lat_avg = C.e4e_inputs(2)
    img, dlat = C.fs_inputs(2)
    hp = HairFastHotPath(args, state, C.params_from_shapes("e4e", E.e4e_param_shapes()),
                         C.params_from_shapes("fs", E.fs_param_shapes()), lat_avg, dlat)
    assert not any(p.requires_grad for p in hp.net.generator.parameters())
    assert hp.net.layer_num == 18 and hp.net.S_index == 7
    torch.manual_seed(0)
    B3 = 3
    images_1024 = torch.cat([img, img[:1]], 0).to(dev)
    images_256 = torch.cat([x256, x256[:1]], 0).to(dev)
    z = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
    out = hp.swap_schedule(images_1024, images_256, x256.to(dev), z(1, 512, 32, 32), z(1, 512, 64, 64), z(1, 18, 512),
                           z(1, 18, 512), z(1, 18, 512))
    assert out["W"].shape == (B3, 18, 512) and out["S"].shape == (B3, 18, 512)
    assert out["F"].shape == (B3, 512, 32, 32) and out["f_from_w"].shape == (B3, 512, 32, 32)
    assert out["F_sean"].shape == (2, 512, 32, 32)
    for k in ("I_rot_shape", "I_rot_color", "I_blend", "I_final"):
        assert out[k].shape == (1, 3, 1024, 1024) and torch.isfinite(out[k]).all()
    # batch rows 0,1 of the e4e / FS outputs equal the golden-verified batch-2 runs (row 2 repeats row 0)
    assert float((out["W"][2] - out["W"][0]).abs().max()) < 1e-5
    assert float((out["S"][2] - out["S"][0]).abs().max()) < 1e-5


def test_graph_runner_matches_eager():
    """hipGraph capture (hairfastgan_amd/graphs.py) of a generator call with explicit noise and of
    the e4e encoder: replays reproduce the eager results bit for bit, also on new inputs."""
    from hairfastgan_amd.graphs import GraphRunner
    from hairfastgan_amd.stylegan2.model import Generator

    dev = torch.device("cuda:0")
    g = Generator(64, 512, 2).eval()
    shapes = {k: tuple(v.shape) for k, v in g.state_dict().items()}
    g.load_state_dict(C.generator_params(shapes))
    g = g.to(dev)
    lat, nz, _ = C.generator_inputs(64, 2, 0)
    nz = [n.to(dev) for n in nz]
    fn = lambda w: g([w], input_is_latent=True, noise=nz)[0]  # noqa: E731
    runner = GraphRunner(fn, lat.to(dev))
    with torch.inference_mode():
        for scale in (1.0, 0.5):
            w = (lat * scale).to(dev)
            assert torch.equal(runner(w), fn(w))
