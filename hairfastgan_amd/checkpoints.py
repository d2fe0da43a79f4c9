"""The reference's on-disk checkpoints -> the state dicts this backend's modules load.

The reference constructor reads every network from a path that its parser carries or that a module hard-codes, relative
to the working directory (a HairFastGAN checkout with `pretrained_models/` downloaded):

| network | file | key handling | reference |
|---|---|---|---|
| StyleGAN2 generator | args.ckpt | ['g_ema'], ['latent_avg'] | models/Net.py:31-46 (hairfastgan_amd.net.Net) |
| e4e | pretrained_models/encoder4editing/e4e_ffhq_encode.pt | ['state_dict'] entries under `encoder.`, ['latent_avg'] | models/Embedding.py:31, encoder4editing/utils/model_utils.py:17-28, models/psp.py:11-15,41-47,93-104 |
| FS encoder | pretrained_models/FeatureStyleEncoder/143_enc.pth; dlatent_avg = psp_ffhq_encode.pt['latent_avg'] | plain state dict | models/FeatureStyleEncoder/FSencoder.py:27-39, trainer.py:188-195 |
| BiSeNet | pretrained_models/BiSeNet/face_parsing_79999_iter.pth | plain | CtrlHair/external_code/face_parsing/my_parsing_util.py:72-81 |
| SEAN generator | pretrained_models/sean_checkpoints/CelebA-HQ_pretrained/latest_net_G.pth + the 19 median style codes models/sean_codes/styles_test/mean_style_code/median/<label>/ACE.npy | plain (netG's own dict) | models/Alignment.py:29-30, sean_codes/util/util.py:204-210, pix2pix_model.py:268-293,328-329 |
| shape adaptor | pretrained_models/ShapeAdaptor/mask_generator.pth | plain | models/Alignment.py:32-34 |
| RotateModel | args.rotate_checkpoint | ['model_state_dict'] | models/Alignment.py:36-38 |
| ClipBlendingModel | args.blending_checkpoint | ['model_state_dict'] (strict=False), ['clip'] names the tower (default ViT-B/32) | models/Blending.py:24-27 |
| CLIP ViT-B/32 | the blending checkpoint's `clip_model.*` entries when it carries them (they overwrite clip.load's weights in the reference's load_state_dict), else clip.load's download cache ~/.cache/clip/ViT-B-32.pt (a TorchScript archive) | `visual.*` | models/Encoders.py:79; requirements.txt:6 |
| PostProcessModel | args.pp_checkpoint ['model_state_dict'] + pretrained_models/PostProcess/latent_avg.pt | | models/Blending.py:29-30, models/Encoders.py:112 |

A file that does not exist raises FileNotFoundError naming it and the network it holds: nothing here ever falls back to
randomly initialised weights, and nothing is downloaded (the reference's gdown / clip downloads need a network).

`root`: directory the relative paths are resolved against (default: the working directory, as in the reference;
`HAIRFAST_PRETRAINED_ROOT` overrides the default).  Absolute paths in `args` are used as they are.
"""
import argparse
import os
import pickle
import warnings
from pathlib import Path

import numpy as np
import torch

E4E_PATH = "pretrained_models/encoder4editing/e4e_ffhq_encode.pt"                      # models/Embedding.py:31
FS_ENCODER_PATH = "pretrained_models/FeatureStyleEncoder/143_enc.pth"                  # FSencoder.py:27
FS_STYLEGAN_PATH = "pretrained_models/FeatureStyleEncoder/psp_ffhq_encode.pt"          # FSencoder.py:27 (latent_avg only)
BISENET_PATH = "pretrained_models/BiSeNet/face_parsing_79999_iter.pth"                 # my_parsing_util.py:78
SEAN_PATH = "pretrained_models/sean_checkpoints/CelebA-HQ_pretrained/latest_net_G.pth"  # SEAN_OPT, util.py:204-207
SEAN_CODES_DIR = "models/sean_codes/styles_test/mean_style_code/median"                # pix2pix_model.py:274
SHAPE_ADAPTOR_PATH = "pretrained_models/ShapeAdaptor/mask_generator.pth"               # models/Alignment.py:34
PP_LATENT_AVG_PATH = "pretrained_models/PostProcess/latent_avg.pt"                     # models/Encoders.py:112
CLIP_CACHE = "~/.cache/clip"                                                            # clip.load's download_root default
CLIP_FILES = {"ViT-B/32": "ViT-B-32.pt"}                                                # clip._MODELS basename


def resolve(path, root=None):
    p = Path(os.path.expanduser(str(path)))
    if p.is_absolute():
        return p
    root = root if root is not None else os.environ.get("HAIRFAST_PRETRAINED_ROOT")
    return (Path(root) / p) if root else p


def load_file(path, what, root=None):
    """torch.load of one of the reference's checkpoint files on the CPU; FileNotFoundError names the file and its role."""
    p = resolve(path, root)
    if not p.is_file():
        raise FileNotFoundError(
            f"{what}: checkpoint file '{p}' not found (the reference reads it from this path relative to the working "
            f"directory; nothing is downloaded here and no network is left randomly initialised - pass the state dict to "
            f"HairFast(...) or set HAIRFAST_PRETRAINED_ROOT / the parser argument)")
    try:  # tensors, dicts, lists, numbers - and the argparse.Namespace some training scripts store
        with torch.serialization.safe_globals([argparse.Namespace]):
            try:  # zip-format files are mapped, not read: a tensor's pages are touched once, by load_state_dict's copy
                return torch.load(str(p), map_location="cpu", weights_only=True, mmap=True)
            except (RuntimeError, ValueError):  # torch < 1.6 legacy (non-zip) files cannot be mapped
                return torch.load(str(p), map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as e:
        # The reference's plain torch.load (torch 1.13) unpickles arbitrary objects - i.e. runs arbitrary code from the file.
        # Not by default here: only with an explicit opt-in for files the user trusts.
        if os.environ.get("HAIRFAST_UNSAFE_LOAD", "0") in ("", "0"):
            raise pickle.UnpicklingError(
                f"{what}: '{p}' does not load with torch.load(weights_only=True): {e}  The file pickles objects beyond tensors / "
                f"containers / argparse.Namespace; loading it would execute code from the file.  If you trust it, set "
                f"HAIRFAST_UNSAFE_LOAD=1 (full unpickling, what the reference's torch.load does), or re-save its state dicts.") from e
        warnings.warn(f"{p}: HAIRFAST_UNSAFE_LOAD=1 - full unpickling (weights_only=False) as the reference does")
        return torch.load(str(p), map_location="cpu", weights_only=False)


def _entry(obj, key, path, what):
    if not isinstance(obj, dict) or key not in obj:
        have = sorted(obj)[:8] if isinstance(obj, dict) else type(obj).__name__
        raise KeyError(f"{what}: '{path}' has no entry '{key}' (found {have})")
    return obj[key]


def e4e(root=None, path=E4E_PATH):
    """-> (Encoder4Editing state dict, latent_avg [18,512]).  get_keys(ckpt, 'encoder') of psp.py:11-15 and
    `__load_latent_avg` (psp.py:93-104; the e4e checkpoint carries `latent_avg`, the 10 000-sample fallback needs the
    decoder and is not reproduced: a checkpoint without the entry raises)."""
    ckpt = load_file(path, "e4e encoder (models/Embedding.py:31)", root)
    sd = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt
    name = "encoder"
    state = {k[len(name) + 1:]: v for k, v in sd.items() if k[:len(name)] == name}
    if not state:
        raise KeyError(f"e4e encoder: '{path}' holds no 'encoder.*' entries")
    return state, _entry(ckpt, "latent_avg", path, "e4e encoder")


def fs_encoder(root=None, path=FS_ENCODER_PATH, stylegan_path=FS_STYLEGAN_PATH):
    """-> (fs_encoder_v2 state dict, dlatent_avg).  FSencoder.py:38 loads 143_enc.pth into trainer.enc; trainer.py:192
    takes dlatent_avg from the pSp checkpoint's 'latent_avg' (its StyleGAN copy there only serves the discarded
    reconstruction, trainer.py:295 - this backend runs that forward, when asked to, on Net's generator)."""
    state = load_file(path, "FeatureStyle encoder (FSencoder.py:38)", root)
    avg = _entry(load_file(stylegan_path, "FeatureStyle encoder's dlatent_avg (trainer.py:188-192)", root), "latent_avg",
                 stylegan_path, "FeatureStyle encoder")
    return state, avg


def bisenet(root=None, path=BISENET_PATH):
    return load_file(path, "BiSeNet face parsing (my_parsing_util.py:77-79)", root)


def sean(root=None, path=SEAN_PATH, codes_dir=SEAN_CODES_DIR):
    """-> (SPADEGenerator state dict, mean_codes [19,512]): load_network (util.py:204-210) and load_average_feature
    (pix2pix_model.py:268-293: the 'ACE' entry of every label's folder)."""
    state = load_file(path, "SEAN generator (models/Alignment.py:29, util.py:204-210)", root)
    codes = []
    for label in range(19):
        f = resolve(Path(codes_dir) / str(label) / "ACE.npy", root)
        if not f.is_file():
            raise FileNotFoundError(f"SEAN median style code of label {label}: '{f}' not found (pix2pix_model.py:268-293 reads "
                                    f"the nineteen <label>/ACE.npy files of the HairFastGAN checkout)")
        code = np.load(str(f))
        if code.shape != (512,):
            raise ValueError(f"{f}: expected a [512] style code, found {code.shape}")
        codes.append(torch.from_numpy(np.ascontiguousarray(code)).float())
    return state, torch.stack(codes)


def shape_adaptor(root=None, path=SHAPE_ADAPTOR_PATH):
    return load_file(path, "CtrlHair shape adaptor (models/Alignment.py:32-34)", root)


def rotate(path, root=None):
    ckpt = load_file(path, "RotateModel (--rotate_checkpoint, models/Alignment.py:36-37)", root)
    return _entry(ckpt, "model_state_dict", path, "RotateModel")


def clip_tower(name="ViT-B/32", root=None):
    """The OpenAI model `clip.load(name)` holds (models/Encoders.py:79): its download cache, a TorchScript archive (or a
    plain state dict saved under the same name).  HAIRFAST_CLIP_WEIGHTS names another file."""
    if name not in CLIP_FILES:
        raise NotImplementedError(f"CLIP model '{name}': only the ViT-B/32 image tower is built (models/Encoders.py:76)")
    path = os.environ.get("HAIRFAST_CLIP_WEIGHTS") or str(Path(CLIP_CACHE) / CLIP_FILES[name])
    p = resolve(path, root)
    if not p.is_file():
        raise FileNotFoundError(
            f"CLIP {name} image tower (models/Encoders.py:79 clip.load): '{p}' not found and the blending checkpoint carries "
            f"no 'clip_model.*' entries; nothing is downloaded here - set HAIRFAST_CLIP_WEIGHTS or pass clip_state")
    try:
        return {k: v for k, v in torch.jit.load(str(p), map_location="cpu").state_dict().items()}
    except RuntimeError:  # not a TorchScript archive: a plain state dict
        obj = load_file(p, f"CLIP {name}")
        return obj.get("state_dict", obj) if isinstance(obj, dict) else obj.state_dict()


def blending(path, root=None, need_tower=True):
    """-> (ClipBlendingModel state dict without the tower, CLIP state dict).  Blending.py:24-26: the tower is
    clip.load(ckpt.get('clip', 'ViT-B/32')) and `load_state_dict(model_state_dict, strict=False)` then overwrites it with
    the checkpoint's own `clip_model.*` entries when it has them."""
    ckpt = load_file(path, "ClipBlendingModel (--blending_checkpoint, models/Blending.py:24)", root)
    sd = _entry(ckpt, "model_state_dict", path, "ClipBlendingModel")
    own = {k: v for k, v in sd.items() if not k.startswith("clip_model.")}
    tower = {k[len("clip_model."):]: v for k, v in sd.items() if k.startswith("clip_model.visual.")}
    if not tower and need_tower:
        tower = clip_tower(ckpt.get("clip", "ViT-B/32"), root)
    return own, tower


def post_process(path, root=None, latent_avg_path=PP_LATENT_AVG_PATH):
    """-> (PostProcessModel state dict, latent_avg).  Blending.py:29-30, Encoders.py:112."""
    ckpt = load_file(path, "PostProcessModel (--pp_checkpoint, models/Blending.py:30)", root)
    avg = load_file(latent_avg_path, "PostProcessModel.latent_avg (models/Encoders.py:112)", root)
    return _entry(ckpt, "model_state_dict", path, "PostProcessModel"), avg
