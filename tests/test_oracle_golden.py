"""CPU tests: the oracle (oracle/ref_stylegan2.py) against the golden vectors that
oracle/make_golden.py produced by running the REAL reference in the build container.
(The 1024^2 generator cases take ~1.5 s per forward on 8 cores.)"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cases as C
from oracle import ref_stylegan2 as O
from oracle import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_report_says_oracle_equals_reference():
    rep = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_vs_reference.json")))
    assert rep["worst"] < 1e-6
    assert all(v == 0.0 for k, v in rep["max_abs_diff"].items() if not k.startswith("upfirdn_loops/"))


def test_synth_is_bit_stable():
    a = synth.unit_uniform("conv1.conv.weight", (1, 512, 512, 3, 3))
    want = np.array([-0.66320264, 1.4886293, 0.8990901, 0.3589508], dtype=np.float32)
    assert np.array_equal(a.flat[:4], want)
    assert synth.fill_value("convs.3.noise.weight", (1,))[0] != 0.0  # zero-init params must be exercised


@pytest.mark.parametrize("name", list(C.UPFIRDN_CASES))
def test_upfirdn2d(golden, name):
    c = C.UPFIRDN_CASES[name]
    y = O.upfirdn2d(C.upfirdn_input(name), C.blur_kernel4(), c["up"], c["down"], c["pad"])
    assert np.array_equal(y.numpy(), golden("upfirdn2d.npz")[name])


@pytest.mark.parametrize("name", [c[0] for c in C.MODCONV_SMALL])
def test_small_modules(golden, name):
    d = C.modconv_small_inputs(name)
    G = golden("modconv_small.npz")
    for up in (False, True):
        P = d[f"P_up{int(up)}"]
        y = O.styled_conv(P, "L", d["x"], d["w"], d[f"noise_up{int(up)}"], up)
        assert np.array_equal(y.numpy(), G[f"{name}_up{int(up)}_styled"])
    y = O.to_rgb(d["P_rgb"], "L", d["x_rgb"], d["w"], d["skip"])
    assert np.array_equal(y.numpy(), G[f"{name}_rgb_skip1"])


def _check_generator(golden, tag, fname, ranges=None):
    size, cm, n_mlp, batches, all_ranges = C.GENERATOR_CASES[tag]
    G = golden(fname)
    shapes = O.generator_param_shapes(size, 512, n_mlp, cm)
    P = C.generator_params(shapes)
    log_size = int(np.log2(size))
    B = batches[0]
    for (s, e) in (ranges or all_ranges):
        cin = shapes[f"convs.{2 * s - 2}.conv.weight"][2] if s > 0 else None
        lat, nz, layer_in = C.generator_inputs(size, B, s, cin)
        y, sk = O.generator_forward(P, lat, nz, layer_in=layer_in, start_layer=s, end_layer=e, log_size=log_size)
        key = f"{tag}_B{B}_r{s}to{e}"
        f = y.reshape(-1)
        step = max(1, f.numel() // 1024)
        assert np.array_equal(f[::step][:1024].numpy(), G[f"{key}_samples"])
        if f"{key}_full" in G:
            assert np.array_equal(y.numpy(), G[f"{key}_full"])


def test_generator64(golden):
    _check_generator(golden, "g64", "generator_64.npz")


def test_generator1024_partial_range(golden):
    # 0->3 is 8 GFLOP: cheap enough for the CPU suite; the full 0->8 forward is exercised
    # by make_golden.py itself and by bench.py's cpu_baseline leg.
    _check_generator(golden, "g1024", "generator_1024.npz", ranges=[(0, 3), (3, 3)])


def test_mapping_network(golden):
    """Row a11: oracle z -> w against the reference's Generator.style, and a 64^2 forward entered
    through z (input_is_latent=False)."""
    for tag, fname in (("g64", "generator_64.npz"), ("g1024", "generator_1024.npz")):
        size, cm, n_mlp, _, _ = C.GENERATOR_CASES[tag]
        P = C.generator_params(O.generator_param_shapes(size, 512, n_mlp, cm))
        w = O.mapping_network(P, C.mapping_inputs(3), n_mlp=n_mlp)
        assert np.array_equal(w.numpy(), golden(fname)[f"{tag}_mapping_w"])
        if tag == "g64":
            _, nz, _ = C.generator_inputs(size, 3, 0)
            y, _ = O.generator_forward(P, w.unsqueeze(1).repeat(1, 10, 1), nz, log_size=6)
            assert np.array_equal(y.numpy(), golden(fname)["g64_from_z_full"])
