"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol
the header declares, the host mirror keeps the reference's names / signatures /
state-dict layout, and the product refuses to run without the GPU (no fallback)."""
import ctypes
import inspect
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    inc = os.path.join(ROOT, "include")
    for fn in os.listdir(inc):
        if fn.endswith(".h"):
            txt = open(os.path.join(inc, fn)).read()
            txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
            names |= set(re.findall(r"\b(hf_\w+)\s*\(", txt))
    return names


def test_library_exports_every_declared_symbol():
    from hairfastgan_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 12
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    assert set(_lib.SIGNATURES) | {"hf_strerror", "hf_abi_version", "hf_modconv_workspace_floats", "hf_conv2d_workspace_floats",
                                    "hf_f16_overflow_count", "hf_conv2d_f16_workspace_floats",
                                    "hf_sample_layernorm_workspace_floats", "hf_conv1x1_f16_workspace_floats",
                                    "hf_modconv3x3_small_workspace_floats"} == declared
    bound = _lib.bind(lib)
    assert bound.hf_abi_version() == 13
    assert bound.hf_strerror(-1) == b"invalid argument"


def test_ops_refuse_cpu_tensors():
    from hairfastgan_amd.stylegan2.op import FusedLeakyReLU, fused_leaky_relu, upfirdn2d

    x = torch.randn(1, 3, 8, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        upfirdn2d(x, torch.ones(4, 4), pad=(1, 1))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        fused_leaky_relu(x, torch.zeros(3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        FusedLeakyReLU(3)(x)
    from hairfastgan_amd.stylegan2.model import Generator

    g = Generator(8, 512, 1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        g([torch.randn(1, 4, 512)], input_is_latent=True)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from hairfastgan_amd import _lib

    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.HairfastLibError, match="no CPU / PyTorch fallback"):
        _lib.load()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "hairfastgan_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f"{f} imports the oracle"
                assert "/root/reference" not in src


def test_generator_surface_matches_reference_contract():
    from hairfastgan_amd.stylegan2.model import Generator
    from oracle.ref_stylegan2 import generator_param_shapes

    g = Generator(1024, 512, 8, channel_multiplier=2)
    sd = g.state_dict()
    want = generator_param_shapes(1024, 512, 8, 2)  # verified == the reference's 171 entries by make_golden.py
    assert len(sd) == 171
    assert list(sd) == list(want)
    assert {k: tuple(v.shape) for k, v in sd.items()} == want
    sig = inspect.signature(g.forward)
    assert list(sig.parameters) == ["styles", "return_latents", "inject_index", "truncation", "truncation_latent",
                                    "input_is_latent", "noise", "randomize_noise", "layer_in", "skip",
                                    "start_layer", "end_layer", "return_rgb"]
    assert sig.parameters["end_layer"].default == 8 and sig.parameters["start_layer"].default == 0
    assert g.n_latent == 18 and g.num_layers == 17 and g.log_size == 10
    from hairfastgan_amd.stylegan2 import op

    assert inspect.signature(op.upfirdn2d).parameters["pad"].default == (0, 0)
    assert inspect.signature(op.fused_leaky_relu).parameters["negative_slope"].default == 0.2


def test_kernels_use_no_scratch_memory():
    """Register spills / stack copies of the kernel-argument struct show up as scratch and
    cost 10-20 % on MI355X (it happened once): every kernel of the library must report
    ScratchSize 0 (hipcc -Rpass-analysis=kernel-resource-usage; cross-compiles without a GPU)."""
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    from concurrent.futures import ThreadPoolExecutor

    csrc = os.path.join(ROOT, "hairfastgan_amd", "csrc")
    files = sorted(f for f in os.listdir(csrc) if f.endswith(".hip") and f != "api.hip")  # api.hip: no kernels

    def remarks(f):  # --cuda-device-only: the remarks come from the device pass, skip the host compile
        return f, subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-c",
                                  os.path.join(csrc, f), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                                 capture_output=True, text=True)

    with ThreadPoolExecutor(max_workers=len(files)) as pool:
        results = list(pool.map(remarks, files))
    for f, out in results:
        assert out.returncode == 0, out.stderr[-2000:]
        pairs = re.findall(r"Function Name: (\S+).*?ScratchSize \[bytes/lane\]: (\d+)", out.stderr, flags=re.S)
        assert pairs, f"{f}: no kernel resource remarks"
        for name, size in pairs:
            # The one exception: the fused upsampling kernel (conv_mfma_h<..., FUSE = true>, csrc/convh.hip) holds 128
            # accumulator registers through a blur epilogue inside a 256-register budget (8 waves per block); what is
            # left spills a few values in the PROLOGUE and reloads them outside the epilogue's store phase (a reload
            # behind a store waits for the store's acknowledgement).  Bounded here so that it cannot grow unnoticed;
            # the pre-split-input form the generator's fast path uses is at 28 bytes.
            fused = "conv_mfma_h" in name and name.endswith("ELb1EEEvN9hf_detail10ConvParamsEPKDF16_S4_")
            limit = (32 if "ELb1ELb1EEEv" in name else 192) if fused else 0
            assert int(size) <= limit, f"{f}: {name}: scratch {size} bytes/lane (limit {limit})"


def test_cached_plans_drop_when_loaded_through_a_parent():
    """Advisor (round 1): derived tensors cached on a module (folded BN, re-laid-out weights) must
    not survive a load_state_dict issued on a PARENT container - nn.Module.load_state_dict does not
    call the children's load_state_dict, only their post-hooks."""
    import argparse

    from hairfastgan_amd.encoders import Encoder4Editing
    from hairfastgan_amd.encoders.fs_encoder import FSEncoder
    from hairfastgan_amd.stylegan2.model import Generator

    holder = torch.nn.Module()
    holder.e4e = Encoder4Editing(50, "ir_se", argparse.Namespace(stylegan_size=1024))
    holder.fs = FSEncoder()
    holder.g = Generator(16, 512, 1)
    unit, head, blk = holder.e4e.body[3], holder.e4e.styles[2], holder.fs.enc.block_2[1]
    conv = holder.g.convs[0].conv
    for m in (holder.e4e, unit, head, blk, holder.fs.enc):
        m._plan = "stale"
    conv._prep, conv._prep_f16 = ("k", 1, 2), ("k", 1, 2)
    holder.g.__dict__["_style_jobs"] = {"x": 1}
    holder.load_state_dict(holder.state_dict())
    for m in (holder.e4e, unit, head, blk, holder.fs.enc):
        assert m._plan is None
    assert conv._prep is None and conv._prep_f16 is None
    assert "_style_jobs" not in holder.g.__dict__


def test_torch_custom_ops_are_registered():
    """north_star: the kernels are 'exposed as torch custom ops'.  torch.ops.hairfast.* exist with schemas,
    propagate shapes on meta tensors (what FakeTensor / torch.compile tracing uses) and have no CPU kernel."""
    import hairfastgan_amd.ops  # noqa: F401

    ops = torch.ops.hairfast
    for name in ("upfirdn2d", "fused_bias_act", "noise_bias_act", "modulated_conv3x3", "modulated_conv3x3_up", "to_rgb", "conv2d"):
        assert hasattr(ops, name)
        assert "Tensor" in str(getattr(ops, name).default._schema)
    x = torch.empty(2, 5, 9, 9, device="meta")
    k = torch.empty(4, 4, device="meta")
    assert ops.upfirdn2d(x, k, 1, 1, 1, 1, 1, 1, 1, 1).shape == (2, 5, 8, 8)      # Blur: pad (1,1)
    assert ops.upfirdn2d(x, k, 2, 2, 1, 1, 2, 1, 2, 1).shape == (2, 5, 18, 18)    # Upsample: pad (2,1)
    assert ops.fused_bias_act(x, torch.empty(5, device="meta"), 0.2, 1.4).shape == x.shape
    wt = torch.empty(9, 5, 8, device="meta")
    s, d = torch.empty(2, 5, device="meta"), torch.empty(2, 8, device="meta")
    assert ops.modulated_conv3x3(x, wt, s, d, None, None, None, 0.2, 1.4).shape == (2, 8, 9, 9)
    assert ops.modulated_conv3x3_up(x, wt, s, d, k, None, None, None, 0.2, 1.4).shape == (2, 8, 18, 18)
    assert ops.to_rgb(x, torch.empty(1, 5, 3, device="meta"), s, None, None, None).shape == (2, 3, 9, 9)
    assert ops.conv2d(x, wt, 3, 2, None, None, None, None, 0, None, 0.0, None).shape == (2, 8, 5, 5)
    with pytest.raises((NotImplementedError, RuntimeError)):  # no CPU kernel registered: the dispatcher refuses
        ops.upfirdn2d(torch.zeros(1, 1, 4, 4), torch.ones(4, 4), 1, 1, 1, 1, 1, 1, 1, 1)


def test_reference_rng_walk_switch(monkeypatch):
    """HAIRFAST_RNG_WALK=reference (advisor, round 2): the generator hands the per-layer draws back to the NoiseInjection
    layers (the reference's order of consumption) and the FS encoder wrapper runs its discarded forward."""
    from hairfastgan_amd import _runtime
    from hairfastgan_amd.stylegan2.model import Generator

    g = Generator(16, 32, 2).eval()
    lat = torch.zeros(1, g.n_latent, 32)
    noise = [None] * g.num_layers
    monkeypatch.delenv("HAIRFAST_RNG_WALK", raising=False)
    assert not _runtime.reference_rng_walk()
    monkeypatch.setenv("HAIRFAST_RNG_WALK", "reference")
    assert _runtime.reference_rng_walk()
    assert g._draw_noise(noise, lat, 0, 8) is noise  # untouched: each layer draws its own


def test_conv_precision_scope_sets_and_restores():
    from hairfastgan_amd import _runtime

    base = _runtime.configured_conv_precision()
    other = "f32" if base != "f32" else "f16"
    with _runtime.conv_precision_scope(other):
        assert _runtime.configured_conv_precision() == other
        with _runtime.conv_precision_scope(None):  # no change
            assert _runtime.configured_conv_precision() == other
    assert _runtime.configured_conv_precision() == base
    with pytest.raises(ValueError):
        _runtime.conv_precision_scope("f8")
    try:
        with _runtime.conv_precision_scope(other):
            raise RuntimeError("x")
    except RuntimeError:
        pass
    assert _runtime.configured_conv_precision() == base  # restored on exceptions too


def test_equal_replacer_matches_the_reference_rule():
    """utils/image_utils.py:14-24: images with equal content collapse to ONE object (later stages test `is`); the 8-bit fast
    path (torch.equal instead of allclose's ten launches) gives the reference's answers, floats keep allclose."""
    from hairfastgan_amd.hair_swap import equal_replacer

    g = torch.Generator().manual_seed(0)
    a = torch.randint(0, 256, (3, 16, 16), dtype=torch.uint8, generator=g)
    c = torch.randint(0, 256, (3, 16, 16), dtype=torch.uint8, generator=g)
    r = equal_replacer([a, a.clone(), c])
    assert r[0] is r[1] and r[2] is not r[0] and r[0].dtype == torch.float32 and float(r[0].max()) <= 1.0
    r = equal_replacer([a, c, c.clone()])
    assert r[1] is r[2] and r[0] is not r[1]
    r = equal_replacer([a, a, a.float() / 255])            # mixed dtypes: allclose on the converted images
    assert r[0] is r[1] and r[2] is r[0]
    f = a.float() / 255
    r = equal_replacer([f, f + 1e-9, c])                   # floats within allclose's tolerance collapse, as in the reference
    assert r[1] is r[0]
    off = a.clone()
    off[0, 0, 0] = (int(off[0, 0, 0]) + 1) % 256           # one 8-bit step apart: not equal, not allclose
    r = equal_replacer([a, off, c])
    assert r[1] is not r[0]
    r = equal_replacer([a, torch.zeros(3, 8, 8, dtype=torch.uint8), c])  # different shapes never compare
    assert r[1] is not r[0]


def test_batch_invariant_plans_do_not_follow_the_batch():
    """hf_set_batch_invariant (the default since round 5; HAIRFAST_DETERMINISTIC=0 opts out): the K partition of a layer is the
    canonical batch-3 launch's whatever the real batch - a launch that leaves the chip empty spreads it over the grid
    (workspace = B x the batch-1 workspace: the same number of slabs per sample), a launch that fills the chip by itself runs it
    inside its blocks (no workspace at all) - whereas whole-launch plans split a batch-48 launch less than a batch-1 launch.
    Host-side planning code of the product library: runs without a GPU."""
    from hairfastgan_amd import _lib, _runtime

    L = _lib.load()
    shapes = [(256, 64, 16, 16, 1), (256, 256, 32, 32, 1), (512, 512, 32, 32, 2)]
    try:
        assert L.hf_set_batch_invariant(1) in (0, 1)
        for k, (cin, cout, h, w, stride) in enumerate(shapes):
            one = L.hf_conv2d_f16_workspace_floats(1, cin, cout, h, w, stride, 1)
            assert one > 0, "the canonical plan must split K for these shapes"
            for b in (2, 3, 8):
                assert L.hf_conv2d_f16_workspace_floats(b, cin, cout, h, w, stride, 1) in (b * one, 0)
            assert L.hf_conv2d_f16_workspace_floats(3, cin, cout, h, w, stride, 1) == 3 * one
            big = L.hf_conv2d_f16_workspace_floats(48, cin, cout, h, w, stride, 1)
            assert big == (48 * one if k == 0 else 0)  # 96 blocks: still spread over the grid; 768 blocks: virtual
            # a split output needs the conv's own epilogue: only the virtual form offers it
            assert L.hf_conv2d_f16_split_output_ok(1, cin, cout, h, w, stride, 3, 1) == 0
            assert L.hf_conv2d_f16_split_output_ok(48, cin, cout, h, w, stride, 3, 1) == (0 if k == 0 else 1)
        g1 = L.hf_conv1x1_f16_workspace_floats(1, 768, 3072, 1, 50, 1, 1)
        assert L.hf_conv1x1_f16_workspace_floats(4, 768, 3072, 1, 50, 1, 1) == 4 * g1 > 0
        assert L.hf_conv1x1_f16_workspace_floats(64, 768, 3072, 1, 50, 1, 1) == 0
        L.hf_set_batch_invariant(0)
        cin, cout, h, w, stride = shapes[1]
        one = L.hf_conv2d_f16_workspace_floats(1, cin, cout, h, w, stride, 1)
        assert L.hf_conv2d_f16_workspace_floats(8, cin, cout, h, w, stride, 1) < 8 * one  # whole-launch plans follow the launch
    finally:
        L.hf_set_batch_invariant(1 if _runtime.batch_invariant() else 0)
    assert _runtime.plan_batch(7) == (_runtime.CANON_BATCH if _runtime.batch_invariant() else 7)


def test_deterministic_env_switch_reaches_the_library():
    """HAIRFAST_DETERMINISTIC (default: on) is read at import and applied to the library by the first `_runtime.lib()` call
    (plans are made inside C calls: the flag must be there before the first launch, not after the first host-side predicate)."""
    import subprocess
    import sys

    code = ("from hairfastgan_amd import _runtime; L = _runtime.lib(); "
            "print(int(_runtime.batch_invariant()), L.hf_set_batch_invariant(1), _runtime.plan_batch(9))")
    for env_val, want in (("1", "1 1 3"), ("0", "0 0 9"), (None, "1 1 3")):
        env = {k: v for k, v in os.environ.items() if k != "HAIRFAST_DETERMINISTIC"}
        if env_val is not None:
            env["HAIRFAST_DETERMINISTIC"] = env_val
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-1500:]
        assert r.stdout.strip().splitlines()[-1] == want, r.stdout


def test_swap_accepts_the_reference_image_forms(tmp_path):
    """hair_swap.py:79-92: `swap` takes a Tensor, a PIL image, an array or a file path.  PIL images and arrays go through
    `F.to_tensor` (HWC uint8 -> CHW float / 255 on the CPU), a path through `read_image(path, mode=RGB)` (uint8 [3,H,W],
    read once per call); anything else raises the reference's TypeError.  (The conversion is host code: no GPU needed.)"""
    import numpy as np
    from PIL import Image

    from hairfastgan_amd.hair_swap import HairFast

    rng = np.random.default_rng(0)
    arr = rng.integers(0, 256, (16, 12, 3), dtype=np.uint8)
    want = torch.from_numpy(arr).permute(2, 0, 1).to(torch.float32).div(255)  # F.to_tensor
    t = torch.from_numpy(arr).permute(2, 0, 1)
    assert HairFast._as_tensor(t) is t                                         # tensors pass through (uint8 or float)
    got = HairFast._as_tensor(arr)
    assert got.dtype is torch.float32 and got.shape == (3, 16, 12) and torch.equal(got, want)
    pil = Image.fromarray(arr)
    got = HairFast._as_tensor(pil)
    assert got.dtype is torch.float32 and torch.equal(got, want)
    rgba = Image.fromarray(np.dstack([arr, np.full((16, 12), 255, np.uint8)]), mode="RGBA")
    assert HairFast._as_tensor(rgba).shape == (4, 16, 12)                      # to_tensor keeps every band
    grey = HairFast._as_tensor(Image.fromarray(arr[:, :, 0]))
    assert grey.shape == (1, 16, 12) and torch.equal(grey[0], want[0])
    assert torch.equal(HairFast._as_tensor(arr.astype(np.float32)), torch.from_numpy(arr.astype(np.float32)).permute(2, 0, 1))
    # files: PNG (lossless) through PIL as uint8 RGB, also from a palette / grey file; read once per call
    png, pal, npy = tmp_path / "a.png", tmp_path / "p.png", tmp_path / "a.npy"
    pil.save(png)
    pil.convert("P", palette=Image.ADAPTIVE, colors=8).save(pal)
    np.save(npy, arr)
    cache = {}
    for path in (png, str(png)):
        got = HairFast._as_tensor(path, cache)
        assert got.dtype is torch.uint8 and torch.equal(got, t)
    assert HairFast._as_tensor(png, cache) is HairFast._as_tensor(png, cache) and len(cache) == 2
    got = HairFast._as_tensor(pal)
    assert got.dtype is torch.uint8 and got.shape == (3, 16, 12)               # mode=RGB: palettes are expanded
    assert torch.equal(HairFast._as_tensor(npy), t)
    with pytest.raises(TypeError, match="Unsupported image format"):
        HairFast._as_tensor([1, 2, 3])
    with pytest.raises(FileNotFoundError):
        HairFast._as_tensor(tmp_path / "missing.png")


def test_equal_replacer_many_equals_the_per_triple_form():
    """`equal_replacer_many` (one host synchronisation per batched pass instead of one per image pair): the same objects are
    identified as `equal_replacer` (utils/image_utils.py:14-24) identifies them triple by triple."""
    from hairfastgan_amd.hair_swap import equal_replacer, equal_replacer_many

    g = torch.Generator().manual_seed(0)
    for shape in ((3, 8, 8), (3, 8, 12)):  # rows compared eight bytes at a time / byte by byte
        mk = lambda: torch.randint(0, 256, shape, dtype=torch.uint8, generator=g)  # noqa: E731
        a, b, c = mk(), mk(), mk()
        d = c.clone()
        d[2, 7, -1] ^= 1  # one bit of the last byte
        triples = [[a, b, c], [a.clone(), a.clone(), c.clone()], [b.clone(), c.clone(), c.clone()], [a.clone(), b.clone(), a.clone()],
                   [c.clone(), c.clone(), c.clone()], [c.clone(), d, c.clone()]]
        for forced in (False, True):  # the per-triple fallback (CPU tensors) and the batched comparison
            got = equal_replacer_many(triples, _any_device=forced)
            for tr, out in zip(triples, got):
                ref = equal_replacer(list(tr))
                same = lambda ims: [ims[i] is ims[j] for i in range(3) for j in range(3)]  # noqa: E731
                assert same(out) == same(ref)
                assert all(torch.equal(o, r) for o, r in zip(out, ref)) and out[0].dtype is torch.float32
    assert len(equal_replacer_many([[a, b, b]], _any_device=True)) == 1  # a repeated object: per-triple form


def test_batched_stage_glue_equals_the_glue_of_each_triple_alone():
    """Alignment.align_images_batch / shape_modules and Blending.blend_images_batch build the masks, the three interpolations of F
    and the blending encoder's inputs for all triples with whole-batch launches: every triple gets the values - bit for bit -
    its own call produces.  The networks in between are per-sample stand-ins here (the GPU suite runs the real ones,
    tests/test_gpu_schedule.py); this test is about the index bookkeeping of the batched form."""
    import types

    from hairfastgan_amd import hair_swap as hs

    g = torch.Generator().manual_seed(5)
    T, hw = 4, 64
    rnd = lambda *shape: torch.randn(*shape, generator=g)  # noqa: E731
    parse = lambda: torch.randint(10, 16, (1, 1, hw, hw), generator=g).float()  # noqa: E731

    def entry():
        im = rnd(1, 3, hw, hw)
        return {"W": rnd(1, 18, 8), "F": rnd(1, 8, 32, 32), "S": rnd(1, 18, 8), "mask": parse(), "image_256": im, "image_norm_256": im * 2 - 1}

    name_to_embed, rotated, triples = {}, {}, []
    for t in range(T):
        face, shape = entry(), entry()
        color = shape if t == 1 else entry()          # triple 1: shape is color
        if t == 2:
            shape = face                              # triple 2: the face is its own shape (no SEAN, no mixing of F)
        for n, e in zip(("face", "shape", "color"), (face, shape, color)):
            name_to_embed[(t, n)] = e
        for n in ("shape", "color"):
            if name_to_embed[(t, n)]["image_256"] is not face["image_256"]:
                rotated[((t, "face"), (t, n))] = rotated[((t, "face"), (t, "shape"))] if n == "color" and color is shape else (None, None, parse())
        triples.append(t)

    per_sample = lambda x: x.flatten(1).mean(1)  # noqa: E731
    de = types.SimpleNamespace(mask=lambda m: (m * 0.5 + 0.125, m * 0.25), hair_from_mask=lambda m: ((m == 13).float() * 0.5 + 0.25, (m == 13).float() * 0.75))
    stages = types.SimpleNamespace(
        sean_inpaint_pairs=lambda im, lab, tgt: im + 0.01 * lab + 0.001 * tgt.repeat_interleave(2, 0),
        blend=lambda s1, s3, a, b: s1 * per_sample(a).reshape(-1, 1, 1) + s3 * per_sample(b).reshape(-1, 1, 1))
    align = hs.Alignment.__new__(hs.Alignment)
    torch.nn.Module.__init__(align)
    align.stages, align.dilate_erosion = stages, de
    align.latent_encoder = lambda ims: {"F": torch.stack([per_sample(i[None]).reshape(1, 1, 1).expand(8, 32, 32) * 1.0 for i in ims])}
    blend = hs.Blending.__new__(hs.Blending)
    torch.nn.Module.__init__(blend)
    blend.stages, blend.dilate_erosion = stages, de
    blend.downsample_256 = lambda x: x * 0.5
    blend.post_process = lambda a, b: (per_sample(a).reshape(-1, 1, 1).expand(-1, 18, 8) + per_sample(b).reshape(-1, 1, 1), per_sample(a).reshape(-1, 1, 1, 1) + b[:, :1])
    gen = lambda styles, layer_in=None, **kw: (torch.sin(1e3 * (per_sample(styles[0]) + per_sample(layer_in))).reshape(-1, 1, 1, 1) * torch.linspace(0.1, 1, 192).reshape(1, 3, 8, 8), None)  # noqa: E731
    blend.net = types.SimpleNamespace(generator=gen)

    def run(ts):
        same = [name_to_embed[(t, "shape")] is name_to_embed[(t, "color")] for t in ts]
        a_shape = align.align_images_batch([((t, "face"), (t, "shape")) for t in ts], name_to_embed, rotated=rotated)
        targets = iter(align.shape_modules([((t, "face"), (t, "color")) for t, sm in zip(ts, same) if not sm], name_to_embed, rotated=rotated))
        a_color = [a_shape[j] if sm else next(targets) for j, sm in enumerate(same)]
        images = blend.blend_images_batch(a_shape, a_color, name_to_embed, [tuple((t, n) for n in ("face", "shape", "color")) for t in ts])
        return a_shape, a_color, images

    b_shape, b_color, b_images = run(triples)
    for j, t in enumerate(triples):
        s_shape, s_color, s_images = run([t])
        one = align.shape_module((t, "face"), (t, "color"), name_to_embed, only_target=True, rotated=rotated)
        assert torch.equal(b_color[j]["HM_X"], one["HM_X"]) and torch.equal(b_color[j]["HM_X"], s_color[0]["HM_X"])
        assert torch.equal(b_shape[j]["HM_X"], s_shape[0]["HM_X"]) and torch.equal(b_shape[j]["latent_F_align"], s_shape[0]["latent_F_align"])
        assert torch.equal(b_images[j], s_images[0])
    assert not torch.equal(b_images[0], b_images[3])
