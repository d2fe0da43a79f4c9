#!/usr/bin/env python3
"""Generate tests/golden/pipeline.npz: ONE complete hair swap run through the REFERENCE's own stage classes on CPU.

TEST INFRASTRUCTURE ONLY (build container: needs /root/reference; never on the GPU box).  Pins the orchestration that
hairfastgan_amd/hair_swap.py re-expresses in batched form - `Embedding.embedding_images` (models/Embedding.py:56-117),
`Alignment.align_images` / `shape_module` (models/Alignment.py:43-181), `Blending.blend_images` (models/Blending.py:36-82)
in the order of `HairFast.__swap_from_tensors` (hair_swap.py:38-61) - by instantiating the reference classes with
`__new__` (their constructors read checkpoint files), attaching the reference's own sub-networks (StyleGAN2 Generator,
Encoder4Editing, fs_encoder_v2, BiSeNet, RotateModel, the CtrlHair mask generator, Pix2PixModel/SPADEGenerator,
ClipBlendingModel, PostProcessModel) filled with the synthetic parameters of oracle.cases, and calling their real methods.

What is not the reference's code, and why:
  * `FSencoder.get_trainer(...).test` (models/FeatureStyleEncoder/trainer.py:357-365, 273-297): the Trainer imports
    face_alignment / lpips and needs CUDA; the four lines around the reference's fs_encoder_v2 (two bilinear 0.5x
    downscales, + dlatent_avg) are restated, as in make_golden.py section (v).  The generator forward whose result the
    reference discards (trainer.py:295) is not run: with deterministic noise it has no observable effect.
  * `clip.load("ViT-B/32")`: un-vendored dependency -> an object whose encode_image is oracle/ref_clip.py (parity unpinned).
  * Randomness is replaced by formulas on both sides: the generator runs with randomize_noise=False (its `noises.*`
    buffers), ACE's `torch.randn(..., device='cuda')` (normalization.py:106) returns oracle.cases.pipeline_sean_noise(k).
  * load_average_feature's .npy files -> oracle.cases.sean_mean_codes().

Stored: every generator call's inputs (latents in full, F tensors as channel subsets + samples), the masks of every parse
and of the shape adaptor (uint8 / packed bits), SEAN's outputs, and the final image (crops, samples, statistics).
"""
import argparse
import functools
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    args = ap.parse_args()
    torch.set_grad_enabled(False)

    from oracle import cases as C
    from oracle import make_golden as MG
    from oracle import ref_clip as RC
    from oracle import ref_encoders as E
    from oracle import ref_postprocess as PP
    from oracle import ref_sean as SN
    from oracle import ref_shape_adaptor as SA
    from oracle import ref_stylegan2 as O

    ref_model, _ref_op = MG.import_reference()
    _t = types

    class _Normalize:  # torchvision.transforms.Normalize (library stand-in: this container has no torchvision)
        def __init__(self, mean, std):
            self.mean, self.std = torch.tensor(mean).view(1, 3, 1, 1), torch.tensor(std).view(1, 3, 1, 1)

        def __call__(self, x):
            return (x - self.mean) / self.std

    class _Id:
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            return x

    class _AttrDict(dict):  # addict.Dict
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            for k_, v_ in list(self.items()):
                if isinstance(v_, dict) and not isinstance(v_, _AttrDict):
                    self[k_] = _AttrDict(v_)

        __getattr__ = dict.get
        __setattr__ = dict.__setitem__

    tvt = sys.modules["torchvision.transforms"]
    for nm in ("Resize", "ToTensor"):
        setattr(tvt, nm, _Id)
    tvt.Normalize = _Normalize
    tvt.Compose = lambda ts: (lambda x, _ts=ts: functools.reduce(lambda v, f: f(v), _ts, x))
    tvt.functional = _t.ModuleType("torchvision.transforms.functional")
    tvt.functional.resize = lambda img, size, interpolation=None: (
        img if tuple(img.shape[-2:]) == tuple(size) else torch.nn.functional.interpolate(img.float(), size=size, mode="nearest").to(img.dtype))
    tvt.InterpolationMode = _t.SimpleNamespace(NEAREST="nearest")
    sys.modules["torchvision.transforms.functional"] = tvt.functional
    tvt.transforms = tvt
    for m in ("torchvision.models", "torchvision.utils", "gdown", "clip", "cv2", "dill", "PIL", "PIL.Image", "addict"):
        if m not in sys.modules:
            try:
                __import__(m)
            except Exception:
                sys.modules[m] = _t.ModuleType(m)
    sys.modules["torchvision.utils"].save_image = lambda *a, **k: None
    sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    sys.modules["addict"].Dict = _AttrDict
    fs_pkg = _t.ModuleType("models.FeatureStyleEncoder")       # its __init__ pulls the Trainer (face_alignment, lpips, CUDA)
    fs_pkg.FSencoder = _t.SimpleNamespace(get_trainer=None)
    fs_pkg.__path__ = [os.path.join(REF, "models", "FeatureStyleEncoder")]
    sys.modules["models.FeatureStyleEncoder"] = fs_pkg
    sys.modules["models.FeatureStyleEncoder.FSencoder"] = fs_pkg.FSencoder
    import torch.utils.model_zoo as _mz

    Pbis = C.pipeline_bisenet_params()
    _mz.load_url = lambda *a, **k: {k_[len("cp.resnet."):]: v_ for k_, v_ in Pbis.items() if k_.startswith("cp.resnet.")}

    import models.Alignment as ref_al
    import models.Blending as ref_bl
    import models.Embedding as ref_emb
    from models import Encoders as ref_enc
    from models import Net as ref_net
    from models.CtrlHair.external_code.face_parsing import model as ref_bs
    from models.CtrlHair.external_code.face_parsing import my_parsing_util as ref_pu
    from models.CtrlHair.shape_branch.config import cfg as ref_cfg
    from models.CtrlHair.shape_branch.model import Generator as RefMaskGenerator
    from models.encoder4editing.models.encoders import psp_encoders as ref_psp
    from models.sean_codes.models import pix2pix_model as ref_pm
    from models.sean_codes.models.networks import normalization as ref_norm
    from models.sean_codes.models.networks.generator import SPADEGenerator
    from utils.bicubic import BicubicDownSample as RefBicubic
    from utils.image_utils import DilateErosion as RefDilateErosion

    sys.path.insert(0, os.path.join(REF, "models", "FeatureStyleEncoder"))
    from arcface import iresnet as ref_iresnet
    from nets.feature_style_encoder import fs_encoder_v2
    import tempfile

    opts = argparse.Namespace(device="cpu", batch_size=3, mixing=0.95, smooth=5, save_all=False, size=1024, latent=512,
                              n_mlp=8, channel_multiplier=2)

    # ---- Net (models/Net.py:20-46) around the reference Generator --------------------------------------------------------
    gen = ref_model.Generator(1024, 512, 8, channel_multiplier=2).eval()
    Pg = C.generator_params({k: tuple(v.shape) for k, v in gen.state_dict().items()})
    gen.load_state_dict(Pg)
    calls = []
    gen_forward = gen.forward

    def recorded_forward(styles, **kw):
        calls.append({"start": kw.get("start_layer", 0), "end": kw.get("end_layer", 8), "latent": styles[0].clone(),
                      "layer_in": None if kw.get("layer_in") is None else kw["layer_in"].clone()})
        out = gen_forward(styles, randomize_noise=False, **kw)
        calls[-1]["out"] = out[0]
        return out

    gen.forward = recorded_forward
    net = ref_net.Net.__new__(ref_net.Net)
    torch.nn.Module.__init__(net)
    net.opts, net.generator, net.latent_avg = opts, gen, torch.zeros(512)

    # ---- BiSeNet singleton (my_parsing_util.py:66-86) ---------------------------------------------------------------------
    bise = ref_bs.BiSeNet(n_classes=19).eval()
    bise.load_state_dict(Pbis)
    ref_pu.FaceParsing.bise_net = bise
    parses = []
    seg_fn = ref_net.get_segmentation

    def recorded_segmentation(img, resize=True):
        m = seg_fn(img, resize=resize)
        parses.append(m.clone())
        return m

    ref_emb.get_segmentation = recorded_segmentation
    ref_al.get_segmentation = recorded_segmentation

    # ---- Embedding (models/Embedding.py:22-37) ---------------------------------------------------------------------------
    e4e_net = ref_psp.Encoder4Editing(50, "ir_se", argparse.Namespace(stylegan_size=1024)).eval()
    Pe = C.params_from_shapes("e4e", E.e4e_param_shapes())
    e4e_net.load_state_dict(Pe)
    _, e4e_latent_avg = C.e4e_inputs(2)
    tmp = tempfile.mktemp()
    torch.save(ref_iresnet.iresnet50().state_dict(), tmp)
    fs = fs_encoder_v2(n_styles=18, opts=argparse.Namespace(arcface_model_path=tmp), residual=False, use_coeff=False,
                       resnet_layer=[4, 5, 6], stride=(2, 2)).eval()
    Pf = C.params_from_shapes("fs", E.fs_param_shapes())
    fs.load_state_dict(Pf)
    _, dlat = C.fs_inputs(2)

    class _FSTrainer:  # Trainer.test / get_image (trainer.py:357-365, 273-297, 61-64)
        def test(self, img=None, return_latent=True):
            x = img
            for _ in range(2):
                x = torch.nn.functional.interpolate(x, scale_factor=0.5, mode="bilinear")
            w_recon, fea = fs(x)
            return [img, None, w_recon + dlat, fea]

    embed = ref_emb.Embedding.__new__(ref_emb.Embedding)
    torch.nn.Module.__init__(embed)
    embed.opts, embed.net, embed.encoder = opts, net, _FSTrainer()
    embed.e4e = argparse.Namespace(encoder=e4e_net, opts=argparse.Namespace(start_from_latent_avg=True), latent_avg=e4e_latent_avg)
    embed.normalize = _Normalize((0.5, 0.5, 0.5), (0.5, 0.5, 0.5))
    embed.to_bisenet = _Normalize((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))
    embed.downsample_512, embed.downsample_256 = RefBicubic(factor=2, cuda=False), RefBicubic(factor=4, cuda=False)

    # ---- Alignment (models/Alignment.py:20-41) ---------------------------------------------------------------------------
    sopt = argparse.Namespace(**vars(ref_pm.SEAN_OPT))
    sopt.gpu_ids = []
    sean = ref_pm.Pix2PixModel.__new__(ref_pm.Pix2PixModel)
    torch.nn.Module.__init__(sean)
    sean.opt, sean.FloatTensor, sean.ByteTensor = sopt, torch.FloatTensor, torch.ByteTensor
    sean.netG, sean.netD, sean.netE = SPADEGenerator(sopt), None, None
    sean.eval()
    sean.load_state_dict(C.sean_params())
    mean_codes = C.sean_mean_codes()
    ref_pm.load_average_feature = lambda: {str(i): {"ACE": mean_codes[i].clone()} for i in range(19)}
    ref_al.decode_sean.__globals__["load_average_feature"] = ref_pm.load_average_feature

    class _TorchWithNoise:  # ACE.forward: torch.randn(B, W, H, 1, device='cuda') -> the k-th formula draw
        def __init__(self):
            self.k = 0

        def __getattr__(self, name):
            return getattr(torch, name)

        def randn(self, *shape, device=None):
            r = C.pipeline_sean_noise(self.k, shape[2])     # [1, H, W] natural order
            self.k += 1
            assert shape[0] == 1 and shape[3] == 1
            return r.transpose(1, 2).unsqueeze(-1).contiguous()

    ref_norm.torch = _TorchWithNoise()
    sean_out = []
    dec = ref_pm.decode_sean

    def recorded_decode(model, code, mask):
        g = dec(model, code, mask)
        sean_out.append(g.clone())
        return g

    ref_al.decode_sean = recorded_decode
    mask_gen = RefMaskGenerator(ref_cfg).eval()
    assert {k: tuple(v.shape) for k, v in mask_gen.state_dict().items()} == SA.param_shapes()
    mask_gen.load_state_dict(C.shape_adaptor_params())
    targets = []
    new_shape = ref_al.get_new_shape

    margins = []

    def recorded_new_shape(gen_, face, hair):
        m = new_shape(gen_, face, hair)
        targets.append(m.clone())
        top2 = gen_.forward_decode_by_code(hair, face)[0].log().topk(2, dim=0).values   # softmax output -> log-probabilities
        margins.append((top2[0] - top2[1]).clone())                                     # = top-1 minus top-2 logit
        return m

    ref_al.get_new_shape = recorded_new_shape
    rot = ref_enc.RotateModel().eval()
    rot.load_state_dict(C.params_from_shapes("rotate", PP.rotate_param_shapes()))
    align = ref_al.Alignment.__new__(ref_al.Alignment)
    torch.nn.Module.__init__(align)
    align.opts, align.latent_encoder, align.net = opts, embed.get_e4e_embed, net
    align.sean_model, align.mask_generator, align.rotate_model = sean, mask_gen, rot
    align.dilate_erosion = RefDilateErosion(dilate_erosion=opts.smooth, device="cpu")
    align.to_bisenet = _Normalize((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))

    # ---- Blending (models/Blending.py:16-34) ---------------------------------------------------------------------------
    Pclip = C.clip_params()

    class _Clip(torch.nn.Module):
        def encode_image(self, x):
            return RC.encode_image(Pclip, x)

    ref_enc.clip.load = lambda name, device=None: (_Clip(), None)
    ref_enc.T = tvt
    blend_enc = ref_enc.ClipBlendingModel().eval()
    blend_enc.load_state_dict(C.params_from_shapes("clipblend", PP.clip_blending_param_shapes()), strict=False)
    torch.save(ref_net.iresnet50().state_dict(), tmp)
    pp = ref_enc.PostProcessModel.__new__(ref_enc.PostProcessModel)
    torch.nn.Module.__init__(pp)
    pp.encoder_face = ref_net.FeatureEncoderMult(fs_layers=[9], opts=argparse.Namespace(arcface_model_path=tmp))
    os.remove(tmp)
    pp.to_feature = ref_enc.FeatureiResnet([[1024, 2], [768, 2], [512, 2]])
    pp.to_latent_1 = torch.nn.ModuleList([ref_enc.ModulationModule(18, i == 4) for i in range(5)])
    pp.to_latent_2 = torch.nn.ModuleList([ref_enc.ModulationModule(18, i == 4) for i in range(5)])
    pp.pixelnorm = ref_enc.PixelNorm()
    pp.eval()
    pp_shapes = PP.post_process_param_shapes()
    lat_shape = pp_shapes.pop("latent_avg")
    pp.load_state_dict(C.params_from_shapes("pp", pp_shapes))
    pp.latent_avg = C.params_from_shapes("pp", {"latent_avg": lat_shape})["latent_avg"] * 0.1
    blend = ref_bl.Blending.__new__(ref_bl.Blending)
    torch.nn.Module.__init__(blend)
    blend.opts, blend.net, blend.blending_encoder, blend.post_process = opts, net, blend_enc, pp
    blend.dilate_erosion = RefDilateErosion(dilate_erosion=opts.smooth, device="cpu")
    blend.downsample_256 = RefBicubic(factor=4, cuda=False)

    # ---- the swap: HairFast.__swap_from_tensors (hair_swap.py:38-61) -------------------------------------------------------
    from collections import defaultdict

    import utils.image_utils as ref_iu
    from utils.image_utils import equal_replacer

    class _FWithLongNearest:  # F.interpolate(mode='nearest') of an int64 mask (image_utils.py:38) exists on CUDA only
        def __getattr__(self, name):
            return getattr(torch.nn.functional, name)

        @staticmethod
        def interpolate(x, **kw):
            if x.dtype == torch.int64 and kw.get("mode") == "nearest":
                return torch.nn.functional.interpolate(x.float(), **kw).long()
            return torch.nn.functional.interpolate(x, **kw)

    ref_iu.F = _FWithLongNearest()

    face, shape, color = equal_replacer(list(C.pipeline_images()))
    images_to_name = defaultdict(list)
    for image, name in zip((face, shape, color), ("face", "shape", "color")):
        images_to_name[image].append(name)
    name_to_embed = embed.embedding_images(images_to_name)
    print("embedding done; hair pixels per mask:", [int((name_to_embed[n]["mask"] == 13).sum()) for n in ("face", "shape", "color")], flush=True)
    align_shape = align.align_images("face", "shape", name_to_embed)
    print("align shape done", flush=True)
    align_color = align.shape_module("face", "color", name_to_embed)
    final = blend.blend_images(align_shape, align_color, name_to_embed)
    print("blend done", flush=True)

    g = {}
    for n in ("face", "shape", "color"):
        e = name_to_embed[n]
        g[f"W_{n}"], g[f"S_{n}"] = e["W"][0].numpy(), e["S"][0].numpy()
        g[f"F_{n}_chan16"] = e["F"][0, ::16].numpy().copy()
        g[f"F_{n}_stats"] = MG.stats(e["F"])
        g[f"mask_{n}"] = e["mask"][0, 0].to(torch.uint8).numpy()
        g[f"image_256_{n}_samples"] = MG.strided_samples(e["image_256"], 512)
    # parses in call order: face, shape, color (512^2 inputs, resized), rot(shape) 1024^2, rot(color) 1024^2
    assert len(parses) == 5 and len(targets) == 2 and len(sean_out) == 2, (len(parses), len(targets), len(sean_out))
    g["rot_mask_shape"], g["rot_mask_color"] = parses[3][0, 0].to(torch.uint8).numpy(), parses[4][0, 0].to(torch.uint8).numpy()
    g["target_mask_shape"], g["target_mask_color"] = targets[0].to(torch.uint8).numpy(), targets[1].to(torch.uint8).numpy()
    g["target_margin_shape"], g["target_margin_color"] = margins[0].to(torch.float16).numpy(), margins[1].to(torch.float16).numpy()
    g["HM_X_shape"] = np.packbits(align_shape["HM_X"][0, 0].numpy().astype(np.uint8))
    g["HM_X_color"] = np.packbits(align_color["HM_X"][0, 0].numpy().astype(np.uint8))
    for d in range(2):
        g[f"sean{d}_crop"] = sean_out[d][:, 96:160, 96:160].numpy().copy()
        g[f"sean{d}_samples"] = MG.strided_samples(sean_out[d], 2048)
        g[f"sean{d}_stats"] = MG.stats(sean_out[d])
    g["latent_F_align_chan16"] = align_shape["latent_F_align"][0, ::16].numpy().copy()
    g["latent_F_align_stats"] = MG.stats(align_shape["latent_F_align"])
    # generator calls in order: (3,3) B3, (0,3) B3, (0,8) rot shape, (0,3) B2 [SEAN outputs], (0,8) rot color, (4,8), (5,8)
    sig = [(c["latent"].shape[0], c["start"], c["end"]) for c in calls]
    assert sig == [(3, 3, 3), (3, 0, 3), (1, 0, 8), (2, 0, 3), (1, 0, 8), (1, 4, 8), (1, 5, 8)], sig
    names = ["fs33", "w03", "rot_shape", "sean03", "rot_color", "blend48", "final58"]
    for nm, c in zip(names, calls):
        g[f"call_{nm}_latent"] = c["latent"].numpy()
        if c["layer_in"] is not None:
            g[f"call_{nm}_layer_in_chan16"] = c["layer_in"][:, ::16].numpy().copy()
            g[f"call_{nm}_layer_in_stats"] = MG.stats(c["layer_in"])
        o = c["out"]
        g[f"call_{nm}_out_stats"] = MG.stats(o)
        g[f"call_{nm}_out_samples"] = MG.strided_samples(o, 1024)
    g["final_stats"] = MG.stats(final)
    g["final_samples"] = MG.strided_samples(final, 4096)
    c0 = 512 - 32
    g["final_crop"] = final[:, c0:c0 + 64, c0:c0 + 64].numpy().copy()
    for nm, (sy, sx) in C.edge_crops(1024).items():
        g[f"final_edges_{nm}"] = final[:, sy, sx].numpy().copy()
    os.makedirs(args.out, exist_ok=True)
    np.savez_compressed(os.path.join(args.out, "pipeline.npz"), **g)
    print("saved", os.path.join(args.out, "pipeline.npz"), {k: g[k] for k in g if k.endswith("_stats")})
    ref_norm.torch = torch


if __name__ == "__main__":
    main()
