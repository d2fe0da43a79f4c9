"""CPU tests of the host-side mirror (hairfastgan_amd.stylegan2.model) wired to the
hipsim-interpreted kernels: layer-range semantics, state-dict compatibility and
end-to-end parity with the oracle on a tiny Generator(16)."""
import pytest
import torch

from oracle import cases as C
from oracle import ref_stylegan2 as O


@pytest.fixture()
def sim_backend(simlib, monkeypatch):
    """Route the product modules to the interpreted kernels (test-only monkeypatch; the
    product itself refuses CPU tensors, see test_host_interface.py)."""
    import sys

    import hairfastgan_amd.stylegan2.model as model

    fa = sys.modules["hairfastgan_amd.stylegan2.op.fused_act"]
    up = sys.modules["hairfastgan_amd.stylegan2.op.upfirdn2d"]  # the package attribute is the function
    ops = sys.modules["hairfastgan_amd.ops"]

    for mod in (model, ops):
        monkeypatch.setattr(mod, "lib", lambda: simlib)
        monkeypatch.setattr(mod, "stream", lambda: None)
    for mod in (model, fa, up):
        monkeypatch.setattr(mod, "require_gpu", lambda *a: None)
    # the public ops dispatch through torch.ops.hairfast.* (CUDA kernels only): give the dispatcher CPU kernels that
    # run the interpreted sources for the duration of the test
    import torch as _torch

    cpu_lib = _torch.library.Library("hairfast", "IMPL")
    cpu_lib.impl("fused_bias_act", ops._fused_bias_act, "CPU")
    cpu_lib.impl("upfirdn2d", ops._upfirdn2d, "CPU")
    yield model
    cpu_lib._destroy()


def _build(model, size, n_mlp=2):
    torch.manual_seed(0)
    g = model.Generator(size, 512, n_mlp, channel_multiplier=2).eval()
    shapes = {k: tuple(v.shape) for k, v in g.state_dict().items()}
    assert shapes == O.generator_param_shapes(size, 512, n_mlp, 2)
    assert list(shapes) == list(O.generator_param_shapes(size, 512, n_mlp, 2))
    P = C.generator_params(shapes)
    g.load_state_dict(P)
    return g, P, shapes


@pytest.mark.parametrize("rng", [(0, 2), (0, 1), (1, 2), (2, 2), (0, 0)])
def test_generator16_ranges(sim_backend, rng):
    s, e = rng
    size, B = 16, 1
    g, P, shapes = _build(sim_backend, size)
    cin = shapes[f"convs.{2 * s - 2}.conv.weight"][2] if s > 0 else None
    lat, nz, layer_in = C.generator_inputs(size, B, s, cin)
    with torch.inference_mode():
        y, sk = g([lat], input_is_latent=True, noise=nz, layer_in=layer_in, start_layer=s, end_layer=e)
    yo, sko = O.generator_forward(P, lat, nz, layer_in=layer_in, start_layer=s, end_layer=e, log_size=4)
    assert y.shape == yo.shape
    assert float((y - yo).abs().max()) < 1e-4 * max(1.0, float(yo.abs().max()))
    if sko is None:
        assert sk is None
    else:
        assert float((sk - sko).abs().max()) < 1e-4 * max(1.0, float(sko.abs().max()))


def test_generator_randomize_noise_false_uses_buffers(sim_backend):
    g, P, _ = _build(sim_backend, 8)
    lat, _, _ = C.generator_inputs(8, 1, 0)
    with torch.inference_mode():  # one interpreted forward (CPU suite budget); the (image, None) form is covered by the range tests
        y, lat_out = g([lat], input_is_latent=True, randomize_noise=False, return_latents=True)
    nz = [P[f"noises.noise_{i}"] for i in range(3)]
    yo, _ = O.generator_forward(P, lat, nz, log_size=3)
    assert lat_out is lat
    assert float((y - yo).abs().max()) < 1e-4 * max(1.0, float(yo.abs().max()))


def test_modules_standalone(sim_backend):
    m = sim_backend
    torch.manual_seed(3)
    x = torch.randn(2, 16, 8, 8)
    w = torch.randn(2, 32)
    conv = m.ModulatedConv2d(16, 8, 3, 32)
    ref = O.modulated_conv2d(x, w, conv.weight, conv.modulation.weight, conv.modulation.bias)
    assert float((conv(x, w) - ref).abs().max()) < 2e-5
    convu = m.ModulatedConv2d(16, 8, 3, 32, upsample=True)
    ref = O.modulated_conv2d(x, w, convu.weight, convu.modulation.weight, convu.modulation.bias, upsample=True,
                             blur_kernel=convu.blur.kernel)
    assert float((convu(x, w) - ref).abs().max()) < 2e-5
    ni = m.NoiseInjection()
    ni.weight.data.fill_(0.7)
    nz = torch.randn(2, 1, 8, 8)
    assert float((ni(x, nz) - (x + 0.7 * nz)).abs().max()) < 1e-6
    up = m.Upsample([1, 3, 3, 1])
    assert float((up(x) - O.upfirdn2d(x, up.kernel, up=2, pad=up.pad)).abs().max()) < 1e-5
    bl = m.Blur([1, 3, 3, 1], pad=(2, 1))
    assert float((bl(x) - O.upfirdn2d(x, bl.kernel, pad=(2, 1))).abs().max()) < 1e-5
    # re-preparing after an in-place weight update
    conv.weight.data.mul_(2.0)
    ref = O.modulated_conv2d(x, w, conv.weight, conv.modulation.weight, conv.modulation.bias)
    assert float((conv(x, w) - ref).abs().max()) < 2e-5


@pytest.mark.parametrize("mode,tol", [("f16x3", 2e-5), ("f32", 2e-5), ("f16", 5e-3)])
def test_styled_conv_precision_modes(sim_backend, simlib, mode, tol):
    """StyledConv through the module API in each matrix-core mode (_runtime.set_conv_precision):
    same-resolution and upsampling layers at a shape the fp16 kernels take, asserting which
    kernel family ran (hf_debug_last_path: 5xx = csrc/convh.hip)."""
    from hairfastgan_amd import _runtime

    m = sim_backend
    torch.manual_seed(4)
    # 32 x 32 planes: 3072 pixels in the canonical batch-3 launch the kernel family is chosen for (batch-invariant plans, the
    # default) - above the small-plane forms (fp32 split-K / tap-GEMM up to 2048)
    x = torch.randn(2, 32, 32, 32)
    w = torch.randn(2, 24)
    prev = _runtime.set_conv_precision(mode)
    try:
        for up in (False, True):
            sc = m.StyledConv(32, 64, 3, 24, upsample=up).eval()
            sc.noise.weight.data.fill_(0.3)
            sc.activate.bias.data.normal_()
            oh, ow = (64, 64) if up else (32, 32)
            nz = torch.randn(2, 1, oh, ow)
            with torch.inference_mode():
                y = sc(x, w, noise=nz)
            fam = simlib.hf_debug_last_path() // 100
            assert (fam == 5) == (mode != "f32"), (mode, up, simlib.hf_debug_last_path())
            P = {f"L.{k}": v.detach() for k, v in sc.state_dict().items()}
            ref = O.styled_conv(P, "L", x, w, nz, up)
            assert float((y - ref).abs().max()) < tol * max(1.0, float(ref.abs().max())), (mode, up)
    finally:
        _runtime.set_conv_precision(prev)
