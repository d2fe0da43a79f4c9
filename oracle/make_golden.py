#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference on CPU.

TEST INFRASTRUCTURE ONLY.  Runs only in the build container (needs
/root/reference); never on the GPU box, never from tests/bench/smoke.  It
imports the reference's own modules (models/stylegan2/model.py etc.) with two
import stubs (torchvision is absent; the nvcc JIT `load()` is disabled so the
reference takes its pure-torch CPU branches, op/fused_act.py:86-94 and
op/upfirdn2d.py:146-149), fills them with oracle.synth's closed-form
parameters, and stores inputs-by-formula + small outputs.  Nothing from the
reference is copied: fixtures hold numbers only.

It also reports how far oracle/ref_stylegan2.py is from the reference on every
case (expected: bit-identical or <=1e-6, same ATen kernels).

Usage:  python oracle/make_golden.py [--out tests/golden] [--skip-big]
"""
import argparse
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present: golden vectors can only be made in the build container")
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")

    class ToPILImage:  # only constructed at import time (model.py:13)
        def __call__(self, x):
            return x

    tvt.ToPILImage = ToPILImage
    tv.transforms = tvt
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.transforms", tvt)
    import torch.utils.cpp_extension as ce

    ce.load = lambda *a, **k: None
    sys.path.insert(0, REF)
    import models.stylegan2.model as ref_model
    import models.stylegan2.op as ref_op

    return ref_model, ref_op


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def stats(x):
    x = x.detach().double()
    return np.array([x.mean().item(), x.std().item(), x.min().item(), x.max().item()], dtype=np.float64)


def strided_samples(x, n=1024):
    f = x.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].clone().numpy()


def maxdiff(a, b):
    return float((a.detach().double() - b.detach().double()).abs().max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--skip-big", action="store_true")
    ap.add_argument("--only", default=None,
                    help="comma list of sections to (re)generate (default all): basic,gen64,gen1024,enc_units,encoders,glue,pp,latent,shape,bisenet,sean")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    torch.set_grad_enabled(False)
    torch.manual_seed(0)

    ref_model, ref_op = import_reference()
    from oracle import ref_stylegan2 as O
    from oracle import cases as C

    report = {}
    only = set(args.only.split(",")) if args.only else None
    want = lambda name: only is None or name in only  # noqa: E731
    rep_path = os.path.join(args.out, "oracle_vs_reference.json")
    if only is not None and os.path.exists(rep_path):
        report.update(json.load(open(rep_path))["max_abs_diff"])  # keep the entries of the sections not re-run

    # ---------------- (i) upfirdn2d, modes 1 and 3 (+ one down=2 sanity case) ---
    k4 = C.blur_kernel4()
    g = {}
    for name, c in (C.UPFIRDN_CASES.items() if want("basic") else ()):
        x = C.upfirdn_input(name)
        y = ref_op.upfirdn2d(x, k4, up=c["up"], down=c["down"], pad=c["pad"])
        yo = O.upfirdn2d(x, k4, up=c["up"], down=c["down"], pad=c["pad"])
        report[f"upfirdn/{name}"] = maxdiff(y, yo)
        if x.numel() <= 2000:
            report[f"upfirdn_loops/{name}"] = maxdiff(y, O.upfirdn2d_loops(x, k4, c["up"], c["down"], c["pad"]))
        g[name] = y.numpy()
    if want("basic"):
        np.savez_compressed(os.path.join(args.out, "upfirdn2d.npz"), **g)

    # ---------------- (ii) fused bias + leaky relu -----------------------------
    g = {}
    for name in (C.ACT_CASES if want("basic") else ()):
        x, b = C.act_inputs(name)
        y = ref_op.fused_leaky_relu(x, b)
        report[f"act/{name}"] = maxdiff(y, O.fused_leaky_relu(x, b))
        g[name] = y.numpy()
    if want("basic"):
        np.savez_compressed(os.path.join(args.out, "fused_act.npz"), **g)

    # ---------------- (iii) small modulated conv / styled conv / to_rgb --------
    g = {}
    for name, cin, cout, sdim, B, H, W in (C.MODCONV_SMALL if want("basic") else ()):
        d = C.modconv_small_inputs(name)
        x, w = d["x"], d["w"]
        for up in (False, True):
            m = ref_model.StyledConv(cin, cout, 3, sdim, upsample=up)
            P = d[f"P_up{int(up)}"]
            m.load_state_dict({k_[2:]: v_ for k_, v_ in P.items()})
            nz = d[f"noise_up{int(up)}"]
            y_conv = m.conv(x, w)
            y = m(x, w, noise=nz)
            yo_conv = O.modulated_conv2d(x, w, P["L.conv.weight"], P["L.conv.modulation.weight"],
                                         P["L.conv.modulation.bias"], True, up, P.get("L.conv.blur.kernel"))
            yo = O.styled_conv(P, "L", x, w, nz, up)
            report[f"mc/{name}/up{int(up)}/conv"] = maxdiff(y_conv, yo_conv)
            report[f"mc/{name}/up{int(up)}/styled"] = maxdiff(y, yo)
            g[f"{name}_up{int(up)}_conv"] = y_conv.numpy()
            g[f"{name}_up{int(up)}_styled"] = y.numpy()
        m = ref_model.ToRGB(cout, sdim)
        P = d["P_rgb"]
        m.load_state_dict({k_[2:]: v_ for k_, v_ in P.items()})
        for use_skip in (False, True):
            sk = d["skip"] if use_skip else None
            y = m(d["x_rgb"], w, sk)
            yo = O.to_rgb(P, "L", d["x_rgb"], w, sk)
            report[f"mc/{name}/rgb/skip{int(use_skip)}"] = maxdiff(y, yo)
            g[f"{name}_rgb_skip{int(use_skip)}"] = y.numpy()
    if want("basic"):
        np.savez_compressed(os.path.join(args.out, "modconv_small.npz"), **g)

    # ---------------- (iv) generators -----------------------------------------
    def run_generator(tag):
        size, cm, n_mlp, batches, ranges = C.GENERATOR_CASES[tag]
        gen = ref_model.Generator(size, 512, n_mlp, channel_multiplier=cm).eval()
        shapes = {k_: tuple(v_.shape) for k_, v_ in gen.state_dict().items()}
        mine = O.generator_param_shapes(size, 512, n_mlp, cm)
        assert mine == shapes, "state-dict layout mismatch between oracle and reference"
        assert list(mine) == list(shapes), "state-dict key ORDER mismatch"
        P = C.generator_params(shapes)
        gen.load_state_dict(P)
        log_size = int(np.log2(size))
        out = {}
        for B in batches:
            for (s, e) in ranges:
                cin = shapes[f"convs.{2 * s - 2}.conv.weight"][2] if s > 0 else None
                lat, nz, layer_in = C.generator_inputs(size, B, s, cin)
                inter = {}
                hooks = []
                if (s, e) == (0, log_size - 2) and B == batches[0]:
                    for nm, mod in list(gen.named_modules()):
                        if nm in ("conv1", "to_rgb1") or (nm.count(".") == 1 and nm.split(".")[0] in ("convs", "to_rgbs")):
                            hooks.append(mod.register_forward_hook(
                                lambda _m, _i, o, nm=nm: inter.__setitem__(
                                    nm, np.concatenate([stats(o), strided_samples(o, 16).astype(np.float64)]))))
                y, sk = gen([lat], input_is_latent=True, noise=nz, layer_in=layer_in,
                            start_layer=s, end_layer=e)
                for h in hooks:
                    h.remove()
                yo, sko = O.generator_forward(P, lat, nz, layer_in=layer_in, start_layer=s, end_layer=e,
                                              log_size=log_size)
                key = f"{tag}_B{B}_r{s}to{e}"
                report[f"gen/{key}"] = maxdiff(y, yo)
                if sk is not None:
                    report[f"gen/{key}/skip"] = maxdiff(sk, sko)
                out[f"{key}_stats"] = stats(y)
                out[f"{key}_samples"] = strided_samples(y)
                if y.numel() <= 100_000:
                    out[f"{key}_full"] = y.numpy()
                elif y.shape[1] == 3:
                    c0 = y.shape[-1] // 2 - 32
                    out[f"{key}_crop"] = y[:, :, c0:c0 + 64, c0:c0 + 64].numpy().copy()
                    out[f"{key}_edges"] = np.array([1])
                    for nm, (sy, sx) in C.edge_crops(y.shape[-1]).items():
                        out[f"{key}_edges_{nm}"] = y[:, :, sy, sx].numpy().copy()
                else:  # feature map early exit: every 16th channel, full spatial
                    out[f"{key}_chan16"] = y[:, ::16].numpy().copy()
                if sk is not None:
                    out[f"{key}_skip_stats"] = stats(sk)
                    out[f"{key}_skip_samples"] = strided_samples(sk)
                for nm, v in inter.items():
                    out[f"{key}_inter_{nm}"] = v
                print("done", key, report[f"gen/{key}"], flush=True)
        # mapping network: z -> w, and one forward entered through z
        z = C.mapping_inputs(3)
        w = gen.style(z)
        report[f"gen/{tag}_mapping"] = maxdiff(w, O.mapping_network(P, z, n_mlp=n_mlp))
        out[f"{tag}_mapping_w"] = w.numpy()
        if size <= 64:
            lat, nz, _ = C.generator_inputs(size, 3, 0)
            y, _ = gen([z], input_is_latent=False, noise=nz)
            yo, _ = O.generator_forward(P, O.mapping_network(P, z, n_mlp=n_mlp).unsqueeze(1).repeat(1, 2 * log_size - 2, 1),
                                        nz, log_size=log_size)
            report[f"gen/{tag}_from_z"] = maxdiff(y, yo)
            out[f"{tag}_from_z_full"] = y.numpy()
        return out

    if want("gen64"):
        g = run_generator("g64")
        np.savez_compressed(os.path.join(args.out, "generator_64.npz"), **g)
    if not args.skip_big and want("gen1024"):
        g = run_generator("g1024")
        np.savez_compressed(os.path.join(args.out, "generator_1024.npz"), **g)

    # ---------------- (v) encoders: units, e4e, FeatureStyle encoder --------------------
    from oracle import ref_encoders as E

    tvm = types.ModuleType("torchvision.models")
    tvu = types.ModuleType("torchvision.utils")
    sys.modules.setdefault("torchvision.models", tvm)
    sys.modules.setdefault("torchvision.utils", tvu)
    from models.encoder4editing.models.encoders import helpers as ref_helpers
    from models.encoder4editing.models.encoders import psp_encoders as ref_psp

    sys.path.insert(0, os.path.join(REF, "models", "FeatureStyleEncoder"))
    from arcface import iresnet as ref_iresnet

    g = {}

    def load_unit(mod, P):
        sd = {k_[2:]: v_ for k_, v_ in P.items()}
        missing = [k_ for k_ in mod.state_dict() if k_ not in sd and not k_.endswith("num_batches_tracked")]
        assert not missing, missing
        mod.load_state_dict(sd, strict=False)
        return mod.eval()

    for name, (in_c, depth, stride, B, H, W) in (C.IRSE_UNIT_CASES.items() if want("enc_units") else ()):
        P = C.params_from_shapes(name, C.irse_unit_shapes(in_c, depth))
        x = C.unit_input(name, (B, in_c, H, W))
        y = load_unit(ref_helpers.bottleneck_IR_SE(in_c, depth, stride), P)(x)
        report[f"enc/{name}"] = maxdiff(y, E.ir_se_unit(P, "u", x, in_c, depth, stride))
        g[name] = y.numpy()
    for name, (in_c, planes, stride, B, H, W) in (C.IBASIC_CASES.items() if want("enc_units") else ()):
        P = C.params_from_shapes(name, C.ibasic_shapes(in_c, planes, stride))
        x = C.unit_input(name, (B, in_c, H, W))
        ds = None
        if stride != 1 or in_c != planes:
            ds = torch.nn.Sequential(ref_iresnet.conv1x1(in_c, planes, stride), torch.nn.BatchNorm2d(planes, eps=1e-05))
        y = load_unit(ref_iresnet.IBasicBlock(in_c, planes, stride, ds), P)(x)
        report[f"enc/{name}"] = maxdiff(y, E.ibasic_block(P, "u", x, stride))
        g[name] = y.numpy()
    for name, (c, spatial, B) in (C.STYLE_BLOCK_CASES.items() if want("enc_units") else ()):
        P = C.params_from_shapes(name, C.style_block_shapes(c, spatial))
        x = C.unit_input(name, (B, c, spatial, spatial))
        y = load_unit(ref_psp.GradualStyleBlock(c, c, spatial), P)(x)
        report[f"enc/{name}"] = maxdiff(y, E.gradual_style_block(P, "u", x))
        g[name] = y.numpy()
    if want("enc_units"):
        np.savez_compressed(os.path.join(args.out, "encoder_units.npz"), **g)

    if not args.skip_big and want("encoders"):
        import argparse as _ap
        import tempfile

        g = {}
        e4e = ref_psp.Encoder4Editing(50, "ir_se", _ap.Namespace(stylegan_size=1024)).eval()
        shapes = {k_: tuple(v_.shape) for k_, v_ in e4e.state_dict().items()}
        assert shapes == E.e4e_param_shapes() and list(shapes) == list(E.e4e_param_shapes())
        P = C.params_from_shapes("e4e", shapes)
        e4e.load_state_dict(P)
        x, latent_avg = C.e4e_inputs(2)
        w = e4e(x) + latent_avg.repeat(x.shape[0], 1, 1)  # get_latents, model_utils.py:9-13
        wo, taps = E.e4e_forward(P, x, latent_avg=latent_avg, return_taps=True)
        report["enc/e4e"] = maxdiff(w, wo)
        g["e4e_w"] = w.numpy()
        for nm, tap in zip(("c1", "c2", "c3"), taps):
            g[f"e4e_{nm}_stats"] = stats(tap)
            g[f"e4e_{nm}_samples"] = strided_samples(tap, 256)
        print("done e4e", report["enc/e4e"], flush=True)
        x3, _ = C.e4e_inputs(3)  # the batch HairFast embeds with (Embedding.py:71)
        w3 = e4e(x3) + latent_avg.repeat(3, 1, 1)
        report["enc/e4e_B3"] = maxdiff(w3, E.e4e_forward(P, x3, latent_avg=latent_avg))
        g["e4e_w_B3"] = w3.numpy()
        print("done e4e B3", report["enc/e4e_B3"], flush=True)

        from nets.feature_style_encoder import fs_encoder_v2

        tmp = tempfile.mktemp()
        torch.save(ref_iresnet.iresnet50().state_dict(), tmp)
        fs = fs_encoder_v2(n_styles=18, opts=_ap.Namespace(arcface_model_path=tmp), residual=False, use_coeff=False,
                           resnet_layer=[4, 5, 6], stride=(2, 2)).eval()  # trainer.py:167-168, configs/001.yaml:26
        os.remove(tmp)
        shapes = {k_: tuple(v_.shape) for k_, v_ in fs.state_dict().items()}
        assert shapes == E.fs_param_shapes() and list(shapes) == list(E.fs_param_shapes())
        P = C.params_from_shapes("fs", shapes)
        fs.load_state_dict(P)
        img, dlat = C.fs_inputs(2)
        x256 = img
        for _ in range(2):  # trainer.py:61-64 downscale(x, 2, 'bilinear')
            x256 = torch.nn.functional.interpolate(x256, scale_factor=0.5, mode="bilinear")
        s_ref, content = fs(x256)
        s_ref = s_ref + dlat  # trainer.py:289
        so, co = E.fs_encoder_test(P, img, dlat)
        report["enc/fs_s"] = maxdiff(s_ref, so)
        report["enc/fs_content"] = maxdiff(content, co)
        g["fs_s"] = s_ref.numpy()
        g["fs_content_chan16"] = content[:, ::16].numpy().copy()
        g["fs_content_stats"] = stats(content)
        g["fs_x256_samples"] = strided_samples(x256, 256)
        print("done fs", report["enc/fs_s"], report["enc/fs_content"], flush=True)
        img3, _ = C.fs_inputs(3)
        x3 = img3
        for _ in range(2):
            x3 = torch.nn.functional.interpolate(x3, scale_factor=0.5, mode="bilinear")
        s3, c3 = fs(x3)
        s3 = s3 + dlat
        so3, co3 = E.fs_encoder_test(P, img3, dlat)
        report["enc/fs_s_B3"] = maxdiff(s3, so3)
        report["enc/fs_content_B3"] = maxdiff(c3, co3)
        g["fs_s_B3"] = s3.numpy()
        g["fs_content_chan16_B3"] = c3[:, ::16].numpy().copy()
        print("done fs B3", report["enc/fs_s_B3"], report["enc/fs_content_B3"], flush=True)
        np.savez_compressed(os.path.join(args.out, "encoders.npz"), **g)

    # ---------------- (vi) glue ops + PostProcessModel --------------------
    import types as _types

    tvt_mod = sys.modules["torchvision.transforms"]

    class _Id:  # Compose / Resize / Normalize are only CONSTRUCTED at import time (models/Net.py:12-14, Encoders.py:75)
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            return x

    for nm in ("Compose", "Resize", "Normalize", "ToTensor", "InterpolationMode"):
        if not hasattr(tvt_mod, nm):
            setattr(tvt_mod, nm, _Id)
    tvt_mod.Compose = lambda ts: (lambda x: x)  # transform_to_256 = Resize((256, 256)): the identity on 256^2 inputs
    tvt_mod.functional = _types.ModuleType("torchvision.transforms.functional")
    sys.modules["torchvision.transforms.functional"] = tvt_mod.functional
    tvt_mod.transforms = tvt_mod
    sys.modules["torchvision.utils"].save_image = lambda *a, **k: None
    sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    for m in ("gdown", "clip"):
        sys.modules.setdefault(m, _types.ModuleType(m))
    from utils.bicubic import BicubicDownSample as RefBicubic
    from utils.image_utils import DilateErosion as RefDilateErosion

    # the reference's own stencil classes -> golden vectors (the product's HIP kernels are checked against them in
    # tests/test_sim_parsing.py and tests/test_gpu_schedule.py)
    g = {}
    xg = C.unit_input("glue/bicubic", (2, 3, 64, 64))
    for f in (2, 4):
        g[f"bicubic{f}"] = RefBicubic(factor=f, cuda=False)(xg).numpy()
    mask = (C.unit_input("glue/mask", (3, 1, 48, 48)) > 0.3).float()
    d_ref, e_ref = RefDilateErosion(dilate_erosion=3, device="cpu").mask(mask)
    g["dilate3"], g["erode3"] = d_ref.numpy(), e_ref.numpy()
    mask5 = (C.unit_input("glue/mask5", (2, 1, 256, 256)) > 0.8).float()
    d_ref, e_ref = RefDilateErosion(dilate_erosion=5, device="cpu").mask(mask5)
    g["dilate5"], g["erode5"] = np.packbits(d_ref.numpy().astype(np.uint8)), np.packbits(e_ref.numpy().astype(np.uint8))
    np.savez_compressed(os.path.join(args.out, "glue.npz"), **g)

    if not args.skip_big and want("pp"):
        import argparse as _ap
        import tempfile

        from models import Encoders as ref_enc
        from models import Net as ref_net
        from oracle import ref_postprocess as PP

        tmp = tempfile.mktemp()
        torch.save(ref_net.iresnet50().state_dict(), tmp)
        pp = ref_enc.PostProcessModel.__new__(ref_enc.PostProcessModel)  # the constructor reads checkpoint files
        torch.nn.Module.__init__(pp)
        pp.encoder_face = ref_net.FeatureEncoderMult(fs_layers=[9], opts=_ap.Namespace(arcface_model_path=tmp))
        os.remove(tmp)
        pp.to_feature = ref_enc.FeatureiResnet([[1024, 2], [768, 2], [512, 2]])
        pp.to_latent_1 = torch.nn.ModuleList([ref_enc.ModulationModule(18, i == 4) for i in range(5)])
        pp.to_latent_2 = torch.nn.ModuleList([ref_enc.ModulationModule(18, i == 4) for i in range(5)])
        pp.pixelnorm = ref_enc.PixelNorm()
        pp.eval()
        shapes = {k_: tuple(v_.shape) for k_, v_ in pp.state_dict().items()}
        mine = PP.post_process_param_shapes()
        lat_shape = mine.pop("latent_avg")
        assert shapes == mine and list(shapes) == list(mine), "PostProcess state-dict layout mismatch"
        P = C.params_from_shapes("pp", shapes)
        pp.load_state_dict(P)
        P["latent_avg"] = C.params_from_shapes("pp", {"latent_avg": lat_shape})["latent_avg"] * 0.1
        pp.latent_avg = P["latent_avg"]
        src, tgt = C.pp_inputs()
        s_ref, f_ref = pp(src, tgt)
        s_o, f_o = PP.post_process_forward(P, src, tgt)
        report["pp/s"], report["pp/f"] = maxdiff(s_ref, s_o), maxdiff(f_ref, f_o)
        g = {"pp_s": s_ref.numpy(), "pp_f_chan16": f_ref[:, ::16].numpy().copy(), "pp_f_stats": stats(f_ref),
             "pp_f_samples": strided_samples(f_ref, 1024)}
        # unit-level goldens (small): one ModulationModule, FeatureiResnet-style block at small width
        xm, em = C.unit_input("pp/mod/x", (2, 18, 512)), C.unit_input("pp/mod/e", (2, 18, 512))
        g["pp_mod_mid"] = pp.to_latent_1[0](xm, em).numpy()
        g["pp_mod_last"] = pp.to_latent_1[4](xm, em).numpy()
        report["pp/mod"] = max(maxdiff(torch.from_numpy(g["pp_mod_mid"]), PP.modulation_module(P, "to_latent_1.0", xm, em, 18, False)),
                               maxdiff(torch.from_numpy(g["pp_mod_last"]), PP.modulation_module(P, "to_latent_1.4", xm, em, 18, True)))
        np.savez_compressed(os.path.join(args.out, "postprocess.npz"), **g)
        print("done postprocess", report["pp/s"], report["pp/f"], flush=True)

    # ---------------- (vi-b) RotateModel, ClipBlendingModel around a stand-in CLIP tower (row f4) --------------------
    if want("latent"):
        from models import Encoders as ref_enc
        from oracle import ref_postprocess as PP

        class _Normalize:  # torchvision.transforms.Normalize (the library, stubbed: this container has no torchvision)
            def __init__(self, mean, std):
                self.mean, self.std = torch.tensor(mean).view(1, 3, 1, 1), torch.tensor(std).view(1, 3, 1, 1)

            def __call__(self, x):
                return (x - self.mean) / self.std

        class _FakeClip(torch.nn.Module):  # stands for clip.load("ViT-B/32")[0] (un-vendored dependency)
            def encode_image(self, x):
                return C.fake_clip_embed(x)

        ref_enc.T.Normalize = _Normalize
        ref_enc.T.Compose = lambda ts: (lambda x, _ts=ts: [x := t_(x) for t_ in _ts][-1])
        ref_enc.clip.load = lambda name, device=None: (_FakeClip(), None)
        w_from, w_to, s_face, s_color, img_face, img_color = C.latent_model_inputs()
        rot = ref_enc.RotateModel().eval()
        shapes = {k_: tuple(v_.shape) for k_, v_ in rot.state_dict().items()}
        assert shapes == PP.rotate_param_shapes() and list(shapes) == list(PP.rotate_param_shapes()), "RotateModel layout mismatch"
        Pr = C.params_from_shapes("rotate", shapes)
        rot.load_state_dict(Pr)
        r_ref = rot(w_from, w_to)
        report["latent/rotate"] = maxdiff(r_ref, PP.rotate_model(Pr, w_from, w_to))
        blend = ref_enc.ClipBlendingModel().eval()
        shapes = {k_: tuple(v_.shape) for k_, v_ in blend.state_dict().items() if not k_.startswith("clip_model.")}
        assert shapes == PP.clip_blending_param_shapes() and list(shapes) == list(PP.clip_blending_param_shapes())
        Pb = C.params_from_shapes("clipblend", shapes)
        blend.load_state_dict(Pb, strict=False)
        b_ref = blend(s_face, s_color, img_face, img_color)
        report["latent/clip_blend"] = maxdiff(b_ref, PP.clip_blending(Pb, s_face, s_color, img_face, img_color, C.fake_clip_embed))
        report["latent/clip_input"] = maxdiff(blend.transform(blend.face_pool(img_face) * 0.5 + 0.5), PP.clip_image_input(img_face))
        np.savez_compressed(os.path.join(args.out, "latent_models.npz"), rotate=r_ref.numpy(), clip_blend=b_ref.numpy())
        print("done latent models", report["latent/rotate"], report["latent/clip_blend"], report["latent/clip_input"], flush=True)

    # ---------------- (vi-c) CtrlHair shape adaptor: the mask generator + get_hair_face_code / get_new_shape (row f4) ----
    if not args.skip_big and want("shape"):
        class _AttrDict(dict):  # `addict.Dict` (the library; not installed here): nested dict with attribute access
            def __init__(self, *a, **k):
                super().__init__(*a, **k)
                for k_, v_ in list(self.items()):
                    if isinstance(v_, dict) and not isinstance(v_, _AttrDict):
                        self[k_] = _AttrDict(v_)

            __getattr__ = dict.get
            __setattr__ = dict.__setitem__

        addict_mod = _types.ModuleType("addict")
        addict_mod.Dict = _AttrDict
        sys.modules.setdefault("addict", addict_mod)
        tvt_mod.functional.resize = lambda img, size, interpolation=None: (
            img if tuple(img.shape[-2:]) == tuple(size) else torch.nn.functional.interpolate(img.float(), size=size, mode="nearest").to(img.dtype))
        tvt_mod.InterpolationMode = _types.SimpleNamespace(NEAREST="nearest")
        from models.CtrlHair.shape_branch import shape_util as ref_su
        from models.CtrlHair.shape_branch.config import cfg as ref_cfg
        from models.CtrlHair.shape_branch.model import Generator as RefMaskGenerator
        from oracle import ref_shape_adaptor as SA

        try:
            from models.CtrlHair.shape_branch.solver import get_hair_face_code as ref_codes, get_new_shape as ref_new_shape
        except Exception as e:  # the solver module pulls training-only dependencies: compose its two 6-line functions
            print("solver import failed (", type(e).__name__, e, "): composing get_hair_face_code / get_new_shape from "
                  "shape_util + the model, as solver.py:248-262 do", flush=True)

            def ref_codes(gen, mask):
                mb = mask[None, None].long()
                hair, face = ref_su.split_hair_face(ref_su.mask_label_to_one_hot(mb))
                return gen.forward_face_encoder(face), gen.forward_hair_encoder(hair, testing=True)

            def ref_new_shape(gen, face_code, hair_code):
                return ref_su.mask_one_hot_to_label(gen.forward_decode_by_code(hair_code, face_code))[0]

        gen = RefMaskGenerator(ref_cfg).eval()
        shapes = {k_: tuple(v_.shape) for k_, v_ in gen.state_dict().items()}
        mine = SA.param_shapes()
        assert shapes == mine and list(shapes) == list(mine), "mask generator state-dict layout mismatch"
        Ps = C.shape_adaptor_params()
        gen.load_state_dict(Ps)
        m1, m2 = C.shape_masks()
        g = {}
        labels, worst_logit, worst_code, flips = [], 0.0, 0.0, 0
        logit_all = []
        for b_ in range(m1.shape[0]):  # one pair at a time, as Alignment.py:74-77 calls it
            face_1, _hair_1 = ref_codes(gen, m1[b_, 0].clone())
            _face_2, hair_2 = ref_codes(gen, m2[b_, 0].clone())
            lab = ref_new_shape(gen, face_1, hair_2)
            prob = gen.forward_decode_by_code(hair_2, face_1)  # softmax over the 19 classes
            lab_o, logit_o, fc_o, hc_o = SA.adapt_shape(Ps, m1[b_:b_ + 1], m2[b_:b_ + 1])
            labels.append(lab)
            logit_all.append(logit_o)
            worst_code = max(worst_code, maxdiff(face_1, fc_o), maxdiff(hair_2, hc_o))
            worst_logit = max(worst_logit, maxdiff(prob, torch.softmax(logit_o, dim=1)))
            flips += int((lab != lab_o[0]).sum())
            top2 = logit_o[0].topk(2, dim=0).values
            g[f"margin_{b_}"] = (top2[0] - top2[1]).to(torch.float16).numpy()
            g[f"face_code_{b_}"], g[f"hair_code_{b_}"] = face_1.numpy(), hair_2.numpy()
        logit_o = torch.cat(logit_all)
        report["shape/codes"], report["shape/softmax"], report["shape/label_flips"] = worst_code, worst_logit, float(flips)
        report.pop("shape/log_softmax", None)
        g["labels"] = torch.stack(labels).to(torch.uint8).numpy()
        g["logits_samples"] = strided_samples(logit_o, 2048)
        g["logits_stats"] = stats(logit_o)
        np.savez_compressed(os.path.join(args.out, "shape_adaptor.npz"), **g)
        print("done shape adaptor", worst_code, worst_logit, flips, flush=True)

    # ---------------- (vii) BiSeNet face parsing + label remap --------------------
    if not args.skip_big and want("bisenet"):
        import torch.utils.model_zoo as _mz

        from oracle import ref_bisenet as BS

        shapes_all = BS.bisenet_param_shapes()
        P = C.bisenet_params()
        # Resnet18.__init__ downloads torchvision's resnet18 (resnet.py:79-85): hand it synthetic tensors instead
        _mz.load_url = lambda *a, **k: {k_[len("cp.resnet."):]: v_ for k_, v_ in P.items() if k_.startswith("cp.resnet.")}
        from models.CtrlHair.external_code.face_parsing import model as ref_bs
        from models.CtrlHair.external_code.face_parsing.my_parsing_util import FaceParsing_tensor

        net = ref_bs.BiSeNet(n_classes=19).eval()
        shapes = {k_: tuple(v_.shape) for k_, v_ in net.state_dict().items()}
        assert shapes == shapes_all and list(shapes) == list(shapes_all), "BiSeNet state-dict layout mismatch"
        net.load_state_dict(P)
        g = {}
        for tag, size in (("512", 512), ("320x384", None)):
            x = C.bisenet_input(tag)
            logits = net(x)[0]
            lo = BS.bisenet_logits(P, x)
            report[f"bisenet/logits_{tag}"] = maxdiff(logits, lo)
            parsing = logits.squeeze(0).argmax(0)
            celeba = FaceParsing_tensor.swap_parsing_label_to_celeba_mask(parsing)
            mask_o = BS.get_segmentation(P, x, resize=False)[0, 0]
            report[f"bisenet/mask_{tag}"] = float((celeba != mask_o).sum())
            mask256 = torch.nn.functional.interpolate(celeba[None, None].float(), size=(256, 256), mode="nearest").long()
            report[f"bisenet/mask256_{tag}"] = float((mask256 != BS.get_segmentation(P, x, resize=True)).sum())
            top2 = logits[0].topk(2, dim=0).values
            g[f"logits_stats_{tag}"] = stats(logits)
            g[f"logits_samples_{tag}"] = strided_samples(logits, 2048)
            g[f"mask_{tag}"] = celeba.to(torch.uint8).numpy()
            g[f"mask256_{tag}"] = mask256[0, 0].to(torch.uint8).numpy()
            g[f"margin_{tag}"] = (top2[0] - top2[1]).to(torch.float16).numpy()  # top-1 minus top-2 logit per pixel
        np.savez_compressed(os.path.join(args.out, "bisenet.npz"), **g)
        print("done bisenet", {k_: v_ for k_, v_ in report.items() if k_.startswith("bisenet")}, flush=True)

    # ---------------- (viii) SEAN inpainting: encode_sean / decode_sean around the real SPADEGenerator (row f4) ----------
    if not args.skip_big and want("sean"):
        import copy as _copy

        for m in ("cv2", "dill", "PIL", "PIL.Image"):  # imported by models/sean_codes/util/util.py, unused on this path
            if m not in sys.modules:
                try:
                    __import__(m)
                except Exception:
                    sys.modules[m] = _types.ModuleType(m)
        from models.sean_codes.models import pix2pix_model as ref_pm
        from models.sean_codes.models.networks import normalization as ref_norm
        from models.sean_codes.models.networks.generator import SPADEGenerator
        from oracle import ref_sean as SN

        opt = _copy.copy(ref_pm.SEAN_OPT)
        opt.gpu_ids = []                                              # use_gpu() False: FloatTensor = torch.FloatTensor
        model = ref_pm.Pix2PixModel.__new__(ref_pm.Pix2PixModel)      # the constructor reads checkpoint files
        torch.nn.Module.__init__(model)
        model.opt = opt
        model.FloatTensor, model.ByteTensor = torch.FloatTensor, torch.ByteTensor
        model.netG, model.netD, model.netE = SPADEGenerator(opt), None, None
        model.eval()
        shapes = {k_: tuple(v_.shape) for k_, v_ in model.state_dict().items()}
        mine = SN.sean_param_shapes()
        assert shapes == mine and list(shapes) == list(mine), "SEAN state-dict layout mismatch"
        Ps = C.sean_params()
        model.load_state_dict(Ps)
        mean_codes = C.sean_mean_codes()
        images, labels, target, noise = C.sean_inputs()

        class _TorchWithNoise:  # ACE.forward draws torch.randn(..., device='cuda') (normalization.py:106): explicit draws
            def __init__(self):
                self.queue = []

            def __getattr__(self, name):
                return getattr(torch, name)

            def randn(self, *shape, device=None):
                r = self.queue.pop(0)
                assert tuple(r.shape) == tuple(shape), (tuple(r.shape), shape)
                return r

        shim = _TorchWithNoise()
        ref_norm.torch = shim
        ref_pm.load_average_feature = lambda: {str(i): {"ACE": mean_codes[i].clone()} for i in range(19)}
        codes_ref = ref_pm.encode_sean(model, images.clone(), labels.clone())
        codes_o = SN.encode_sean(Ps, images, labels)
        report["sean/codes"] = maxdiff(codes_ref, codes_o)
        g = {"codes": codes_ref.numpy()}
        gens = []
        for d in range(2):
            shim.queue = [n.clone() for n in noise[d]]
            gen_ref = ref_pm.decode_sean(model, codes_ref[d].unsqueeze(0), target.clone())   # [3,256,256]
            assert not shim.queue
            taps = {}
            gen_o = SN.spade_generator(Ps, SN.one_hot(target), SN.merge_codes(codes_o[d:d + 1], mean_codes), noise[d], taps=taps)
            report[f"sean/decode{d}"] = maxdiff(gen_ref, gen_o[0])
            gens.append(gen_ref)
            g[f"gen{d}_stats"] = stats(gen_ref)
            g[f"gen{d}_samples"] = strided_samples(gen_ref, 2048)
            g[f"gen{d}_crop"] = gen_ref[:, 96:160, 96:160].numpy().copy()
            g[f"gen{d}_corner"] = gen_ref[:, :32, :32].numpy().copy()
            for nm, v in taps.items():  # oracle-side taps (the oracle equals the reference on the outputs above)
                g[f"gen{d}_tap_{nm}"] = np.concatenate([stats(v), strided_samples(v, 64).astype(np.float64)])
        both = SN.sean_inpaint(Ps, images, labels, target, mean_codes, noise[0], noise[1])
        report["sean/inpaint"] = max(maxdiff(both[0], gens[0]), maxdiff(both[1], gens[1]))
        ref_norm.torch = torch
        np.savez_compressed(os.path.join(args.out, "sean.npz"), **g)
        print("done sean", {k_: v_ for k_, v_ in report.items() if k_.startswith("sean")},
              {k_: g[k_] for k_ in g if k_.endswith("_stats")}, flush=True)

    worst = max(report.values())
    with open(rep_path, "w") as f:
        json.dump({"torch": torch.__version__, "max_abs_diff": report, "worst": worst}, f, indent=1, sort_keys=True)
    print(json.dumps(report, indent=1, sort_keys=True))
    print("worst oracle-vs-reference max-abs diff:", worst)


if __name__ == "__main__":
    main()
