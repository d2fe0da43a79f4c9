"""Test infrastructure: writes a `pretrained_models/` tree in the REFERENCE's on-disk layout - file names, wrapper dicts,
key prefixes and decoy entries as the reference's constructors meet them (hair_swap.py:27-37, models/Net.py:31-46,
models/Embedding.py:24-38 -> encoder4editing/utils/model_utils.py:17-28 + models/psp.py:41-47,
FeatureStyleEncoder/FSencoder.py:27-39 + trainer.py:188-201, models/Alignment.py:26-38, models/Blending.py:20-30,
my_parsing_util.py:72-81, sean_codes/models/pix2pix_model.py:268-293) - from in-memory state dicts, and the (cheap or
closed-form) state dicts themselves.  Used by tests/test_checkpoints.py (CPU) and tests/test_gpu_checkpoints.py."""
import os
import zlib

import numpy as np
import torch
from torch import nn

from hairfastgan_amd import checkpoints as CK
from oracle import ref_bisenet as BS
from oracle import ref_clip as CL
from oracle import ref_encoders as E
from oracle import ref_postprocess as PP
from oracle import ref_sean as SN
from oracle import ref_shape_adaptor as SA
from oracle import ref_stylegan2 as O


def all_shapes():
    pp = dict(PP.post_process_param_shapes())
    pp_avg = pp.pop("latent_avg")
    return {
        "generator": O.generator_param_shapes(1024, 512, 8, 2), "e4e": E.e4e_param_shapes(), "fs": E.fs_param_shapes(),
        "bisenet": BS.bisenet_param_shapes(), "sean": SN.sean_param_shapes(prefix=""), "shape": SA.param_shapes(),
        "rotate": PP.rotate_param_shapes(), "blend": PP.clip_blending_param_shapes(), "clip": CL.clip_visual_param_shapes(),
        "pp": pp, "_pp_avg": {"latent_avg": pp_avg},
    }


def cheap_fill(prefix, shapes):
    """Distinguishable, cheap values (a constant per tensor derived from its key + a short ramp): the loader tests compare
    what the modules hold with what the files held, they run no forward."""
    out = {}
    for k, s in shapes.items():
        if k.endswith("num_batches_tracked"):
            out[k] = torch.tensor(zlib.crc32(f"{prefix}.{k}".encode()) % 1000, dtype=torch.int64)
            continue
        base = (zlib.crc32(f"{prefix}.{k}".encode()) % 2003) / 2003.0 + 0.25  # positive: BatchNorm variances, spectral norms
        v = torch.full(tuple(s), base, dtype=torch.float32)
        flat = v.view(-1)
        n = min(flat.numel(), 64)
        flat[:n] += torch.arange(n, dtype=torch.float32) * 1e-3
        out[k] = v
    return out


def cheap_states():
    sh = all_shapes()
    st = {name: cheap_fill(name, shapes) for name, shapes in sh.items() if not name.startswith("_")}
    st["generator_latent_avg"] = torch.full((512,), 0.125)
    st["e4e_latent_avg"] = cheap_fill("e4e_avg", {"a": (18, 512)})["a"]
    st["fs_dlatent_avg"] = cheap_fill("fs_avg", {"a": (18, 512)})["a"]
    st["pp_latent_avg"] = cheap_fill("pp_avg", {"a": (1, 18, 512)})["a"]
    st["sean_mean_codes"] = cheap_fill("sean_codes", {"a": (19, 512)})["a"]
    return st


def _module_tree(sd):
    root = nn.Module()
    for k, v in sd.items():
        m, parts = root, k.split(".")
        for p_ in parts[:-1]:
            if not hasattr(m, p_):
                m.add_module(p_, nn.Module())
            m = getattr(m, p_)
        m.register_parameter(parts[-1], nn.Parameter(v.clone(), requires_grad=False))
    return root


def write_reference_tree(root, st, clip_mode="in_checkpoint"):
    """clip_mode: 'in_checkpoint' (the blending checkpoint carries clip_model.* entries, as a state_dict() of the
    reference's ClipBlendingModel does), 'jit_cache' (a TorchScript archive where clip.load caches its download; returned
    path goes into HAIRFAST_CLIP_WEIGHTS), 'absent'.  Returns the path of the CLIP archive or None."""
    def save(obj, rel):
        path = os.path.join(root, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        torch.save(obj, path)

    decoy = {"decoder.conv1.weight": torch.zeros(3), "latent_avg_not": torch.zeros(1)}
    save({"g_ema": st["generator"], "latent_avg": st["generator_latent_avg"], "g": {"x": torch.zeros(1)}}, "pretrained_models/StyleGAN/ffhq.pt")
    save({"state_dict": {**{"encoder." + k: v for k, v in st["e4e"].items()}, **decoy}, "latent_avg": st["e4e_latent_avg"],
          "opts": {"stylegan_size": 1024, "encoder_type": "Encoder4Editing", "start_from_latent_avg": True}}, CK.E4E_PATH)
    save(st["fs"], CK.FS_ENCODER_PATH)
    save({"state_dict": dict(decoy), "latent_avg": st["fs_dlatent_avg"], "opts": {"output_size": 1024}}, CK.FS_STYLEGAN_PATH)
    save(st["bisenet"], CK.BISENET_PATH)
    save({k[len("netG."):] if k.startswith("netG.") else k: v for k, v in st["sean"].items()}, CK.SEAN_PATH)  # netG's own dict
    for label in range(19):
        d = os.path.join(root, CK.SEAN_CODES_DIR, str(label))
        os.makedirs(d, exist_ok=True)
        np.save(os.path.join(d, "ACE.npy"), st["sean_mean_codes"][label].numpy())
    save(st["shape"], CK.SHAPE_ADAPTOR_PATH)
    save({"model_state_dict": st["rotate"], "epoch": 7, "optimizer_state_dict": {"state": {}, "param_groups": []}},
         "pretrained_models/Rotate/rotate_best.pth")
    blend = dict(st["blend"])
    clip_path = None
    if clip_mode == "in_checkpoint":
        blend.update({"clip_model." + k: v for k, v in st["clip"].items()})
        blend["clip_model.logit_scale"] = torch.tensor(4.6)            # the text tower's entries are ignored
        blend["clip_model.token_embedding.weight"] = torch.zeros(4, 4)
    elif clip_mode == "jit_cache":
        clip_path = os.path.join(root, "clip_cache", "ViT-B-32.pt")
        os.makedirs(os.path.dirname(clip_path), exist_ok=True)
        torch.jit.script(_module_tree({**st["clip"], "logit_scale": torch.tensor(4.6)})).save(clip_path)
    save({"model_state_dict": blend, "clip": "ViT-B/32"}, "pretrained_models/Blending/checkpoint.pth")
    save({"model_state_dict": st["pp"]}, "pretrained_models/PostProcess/pp_model.pth")
    save(st["pp_latent_avg"], CK.PP_LATENT_AVG_PATH)
    return clip_path
