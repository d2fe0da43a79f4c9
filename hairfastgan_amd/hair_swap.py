"""Call surface of the reference's hair_swap.py on the MI355X backend.

`HairFast(args)` / `.swap(face, shape, color, benchmark=False, align=False, seed=None,
exp_name=None)` (hair_swap.py:27-103) with the three stage objects the reference builds -
`Embedding` (models/Embedding.py), `Alignment` (models/Alignment.py), `Blending`
(models/Blending.py) - and `get_parser()` with the reference's defaults (:108-133).

What runs where:

* the HOT PATH of every stage - the e4e and FeatureStyle encoder forwards, the PostProcess encoder
   and all generator calls, with the batch sizes and layer ranges the
  reference issues (Embedding.py:51-53,71-90, Alignment.py:63, Blending.py:62,66,68) - runs on the HIP
  library (hairfastgan_amd.encoders, hairfastgan_amd.stylegan2);
* BiSeNet face parsing + get_segmentation (row f2) likewise (hairfastgan_amd.face_parsing);
* the networks BETWEEN those calls - RotateModel, ClipBlendingModel incl. its CLIP ViT-B/32
  image tower, the CtrlHair shape adaptor, SEAN encode / decode - run natively too (`NativeLatentStages`).  `Stages`
  remains as the injection point for a caller who wants one of the reference's own modules instead
  (INTEGRATION.md shows the binding);
* CHECKPOINTS: like the reference, `HairFast(args)` reads every network from the file its parser carries or the
  reference hard-codes (hairfastgan_amd.checkpoints: the key handling of each file); a missing file raises
  FileNotFoundError naming it.  State dicts passed to the constructor take precedence (tests, benchmarks);
* the stencils on either side of the path (BicubicDownSample, DilateErosion: row f2) are HIP kernels too;
  the remaining glue (normalisation, mask arithmetic, two small F.interpolate calls) is torch elementwise
  code, as in the reference.

Scheduling changes against the reference, none of which changes a per-sample result
(frozen weights, independent samples): the two batch-1 full generator forwards of `Alignment.shape_module`
(Alignment.py:63, once for (face, shape) and once for (face, color)) are issued as ONE batch-2 forward together with
their parses and shape-adaptor calls (`Alignment.rotate_images`); `HairFast.swap_batch` runs several triples as one
batched pass over every hot-path call; for a single swap the three independent branches of the Embedding stage are
enqueued on separate HIP streams.
"""
import argparse
import contextlib
import os
import random
import sys
import time
from collections import defaultdict
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import _marshal as M
from . import checkpoints as ckpt_files
from ._runtime import batch_invariant, batch_invariant_scope, configured_conv_precision, conv_precision_scope, lib, reference_rng_walk, require_gpu, run_guarded, stream
from .encoders import ClipBlendingModel, Encoder4Editing, FSEncoder, PostProcessModel, RotateModel, get_latents
from .face_parsing import BiSeNet, get_segmentation
from .net import Net


def get_parser():
    parser = argparse.ArgumentParser(description="HairFast")
    parser.add_argument("--save_all_dir", type=Path, default=Path("output"))
    parser.add_argument("--size", type=int, default=1024)
    parser.add_argument("--ckpt", type=str, default="pretrained_models/StyleGAN/ffhq.pt")
    parser.add_argument("--channel_multiplier", type=int, default=2)
    parser.add_argument("--latent", type=int, default=512)
    parser.add_argument("--n_mlp", type=int, default=8)
    parser.add_argument("--device", type=str, default="cuda")
    parser.add_argument("--batch_size", type=int, default=3, help="batch size for encoding images")
    parser.add_argument("--save_all", action="store_true")
    parser.add_argument("--mixing", type=float, default=0.95)
    parser.add_argument("--smooth", type=int, default=5)
    parser.add_argument("--rotate_checkpoint", type=str, default="pretrained_models/Rotate/rotate_best.pth")
    parser.add_argument("--blending_checkpoint", type=str, default="pretrained_models/Blending/checkpoint.pth")
    parser.add_argument("--pp_checkpoint", type=str, default="pretrained_models/PostProcess/pp_model.pth")
    return parser


# ---------------------------------------------------------------------------------------------
# glue (torch ops, as in the reference)
# ---------------------------------------------------------------------------------------------
_CONSTS = {}
# bench.py / profiling: a list that receives (label, event) marks at the stage boundaries of a swap (None = off)
STAGE_MARKS = None


def _mark(label):
    if STAGE_MARKS is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        STAGE_MARKS.append((label, ev))


def _const(values, like):
    """A small per-channel constant as a [1,C,1,1] tensor on `like`'s device, created once (a host-to-device copy per
    call would also be illegal inside a hipGraph capture)."""
    key = (tuple(values), like.device, like.dtype)
    if key not in _CONSTS:
        _CONSTS[key] = torch.tensor(values, device=like.device, dtype=like.dtype).view(1, -1, 1, 1)
    return _CONSTS[key]


def _normalize(x, mean, std):
    return (x - _const(mean, x)) / _const(std, x)


class BicubicDownSample(nn.Module):
    """utils/bicubic.py:6-75: separable bicubic (a = -0.5) down-sampling by `factor` with reflect padding
    (hf_bicubic_down_f32; the reference's two strided F.conv2d passes fall to MIOpen's naive kernel on this
    stack: 2 ms per call, a third of a swap)."""

    def __init__(self, factor=4):
        super().__init__()
        self.factor = factor
        size, a = factor * 4, -0.5
        x = (torch.arange(size, dtype=torch.float32) - float(size // 2) + 0.5) / factor
        ax = x.abs()
        k = torch.where(ax <= 1.0, (a + 2.0) * ax ** 3 - (a + 3.0) * ax ** 2 + 1.0,
                        torch.where(ax < 2.0, a * ax ** 3 - 5.0 * a * ax ** 2 + 8.0 * a * ax - 4.0 * a, torch.zeros_like(ax)))
        self.register_buffer("k", k / k.sum(), persistent=False)

    def forward(self, x):
        require_gpu(x)
        if self.k.device != x.device:
            self.k = self.k.to(x.device)
        return M.bicubic_down(lib(), stream(), x, self.k, self.factor)


class DilateErosion:
    """utils/image_utils.py:27-55: `dilate_erosion` rounds of 4-neighbourhood dilation / erosion of BINARY masks
    (hf_dilate_erode_f32: one pass with the equivalent diamond)."""

    def __init__(self, dilate_erosion=5, device="cuda"):
        self.dilate_erosion = dilate_erosion

    def hair_from_mask(self, mask):
        mask = torch.where(mask == 13, torch.ones_like(mask), torch.zeros_like(mask))
        mask = F.interpolate(mask.float(), size=(256, 256), mode="nearest")
        return self.mask(mask)

    def mask(self, mask):
        require_gpu(mask)
        return M.dilate_erode(lib(), stream(), mask.float(), self.dilate_erosion)


def _cat(tensors):
    """torch.cat(tensors, 0) without the copy for a single tensor (one triple: no launch added to the unbatched swap)."""
    tensors = list(tensors)
    return tensors[0] if len(tensors) == 1 else torch.cat(tensors, 0)


def _same_content(a, b, fa, fb):
    """torch.allclose(fa, fb) of utils/image_utils.py:21 without its ten elementwise launches where the answer is exact:
    the same storage, or two 8-bit images (distinct 8-bit values differ by 1/255 after the division, far above allclose's
    1e-5 relative + 1e-8 absolute tolerance: allclose <=> equal)."""
    if a is b or (a.dtype == b.dtype and a.device == b.device and a.data_ptr() == b.data_ptr() and a.stride() == b.stride()):
        return True
    if a.dtype is torch.uint8 and b.dtype is torch.uint8 and a.device == b.device:
        return torch.equal(a, b)
    return torch.allclose(fa, fb)


def equal_replacer(images):
    """utils/image_utils.py:14-24: uint8 -> [0,1]; images with equal content become the same object."""
    raw = list(images)
    images = [im / 255 if im.dtype is torch.uint8 else im for im in raw]
    for i in range(len(images)):
        for j in range(i + 1, len(images)):
            if images[i].shape == images[j].shape and _same_content(raw[i], raw[j], images[i], images[j]):
                images[j] = images[i]
                raw[j] = raw[i]
    return images


def equal_replacer_many(triples, _any_device=False):
    """`equal_replacer` for the T triples of a batched pass with ONE device synchronisation instead of up to 3 T: when every
    image is an 8-bit image of one shape on one device (what `parallel.swap_many` feeds), the 3 T pair comparisons are three
    batched `!=` / `any` reductions over stacked images and one copy of their T x 3 flags to the host.  Anything else
    (float images, mixed shapes / devices, repeated objects) takes the per-triple form.  Same results."""
    flat = [im for tr in triples for im in tr]
    first = flat[0] if flat else None
    batched = (len(triples) > 1 and all(len(tr) == 3 for tr in triples) and first is not None and (first.is_cuda or _any_device)
               and all(torch.is_tensor(im) and im.dtype is torch.uint8 and im.shape == first.shape and im.device == first.device for im in flat)
               and len({id(im) for im in flat}) == len(flat))
    if not batched:
        return [tuple(equal_replacer(list(tr))) for tr in triples]
    cols = [torch.stack([tr[k] for tr in triples]) for k in range(3)]
    # eight bytes per compared element where the rows allow it (the comparison is for equality only)
    wide = [c.view(torch.int64) if c.shape[-1] % 8 == 0 else c for c in cols]
    differ = torch.stack([(wide[i] != wide[j]).flatten(1).any(1) for i, j in ((0, 1), (0, 2), (1, 2))], 1).cpu()  # [T, 3]: the one sync
    scaled = [c / 255 for c in cols]  # three launches for the 3 T divisions; a triple's images are rows of them (same bits)
    out = []
    for t in range(len(triples)):
        images = [scaled[k][t] for k in range(3)]
        d01, d02, d12 = differ[t].tolist()
        if not d01:
            images[1] = images[0]
        if not d02:
            images[2] = images[0]
        elif not d12:
            images[2] = images[1]
        out.append(tuple(images))
    return out


def _to_tensor_like_torchvision(arr):
    """torchvision.transforms.functional.to_tensor for arrays (hair_swap.py:81-82): HWC (or HW) -> CHW; uint8 -> float32 / 255
    ON THE CPU (torch's GPU division is a multiplication by the rounded reciprocal: one ulp away), other dtypes unchanged."""
    if arr.ndim == 2:
        arr = arr[:, :, None]
    if arr.ndim != 3:
        raise ValueError(f"image arrays are HW or HWC, got shape {arr.shape}")
    t = torch.from_numpy(np.array(arr.transpose(2, 0, 1), order="C"))  # a copy: PIL hands out read-only buffers
    return t.to(torch.float32).div(255) if t.dtype is torch.uint8 else t


def _read_image_rgb(path):
    """torchvision.io.read_image(path, mode=ImageReadMode.RGB) (hair_swap.py:85-86): the decoded file as uint8 [3,H,W]."""
    try:
        from PIL import Image
    except ImportError as e:  # pragma: no cover - PIL is in the image
        raise ImportError(f"reading image file {path!r} needs Pillow (pass a tensor, an array or a .npy file otherwise)") from e
    with Image.open(path) as im:
        return torch.from_numpy(np.asarray(im.convert("RGB")).transpose(2, 0, 1).copy())


def set_seed(seed):  # utils/seed.py:8-16
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


# ---------------------------------------------------------------------------------------------
# stages outside the hot path
# ---------------------------------------------------------------------------------------------
class Stages:
    """The networks between the hot-path calls, as named callables.  Every entry documents the
    reference module it stands for and its tensor contract; the defaults raise."""

    def _missing(self, what):
        raise NotImplementedError(
            f"stage '{what}' is outside this backend's scope (SURVEY.md section 8); construct HairFast with "
            f"stages=<object providing {what}()> - the reference's own module (INTEGRATION.md)")

    def rotate(self, w_source_0_6, w_target_0_6):
        """models/Encoders.py:60-72 RotateModel: ([P,6,512], [P,6,512]) -> [P,6,512] (P pairs: 1 or 2 per triple)."""
        self._missing("rotate")

    def shape_adaptor(self, mask_target_pose, mask_hair_source):
        """models/CtrlHair/shape_branch/solver.py:248-262 get_hair_face_code + get_new_shape:
        two long [P,1,256,256] label maps -> long [P,1,256,256] target label maps (P pairs in one call)."""
        self._missing("shape_adaptor")

    def sean_inpaint(self, images_256, labels, target_mask):
        """models/sean_codes/models/pix2pix_model.py:299-325 encode_sean / decode_sean:
        ([2,3,256,256] in [0,1], long [2,1,256,256], long [1,1,256,256]) -> two [3,256,256] images
        normalised to [-1,1] (what Embedding.get_e4e_embed receives, Alignment.py:128-131)."""
        self._missing("sean_inpaint")

    def blend(self, s_face_6_18, s_color_6_18, image_face_masked, image_color_masked):
        """models/Encoders.py:75-103 ClipBlendingModel: ([T,12,512], [T,12,512], [T,3,256,256] x 2) ->
        S_blend[:, 6:18] [T,12,512] (T triples in one call)."""
        self._missing("blend")

    # Not stages any more (built natively): PostProcessModel
    # (hairfastgan_amd.encoders.PostProcessModel, owned by Blending) and BiSeNet face parsing / get_segmentation
    # (hairfastgan_amd.face_parsing, owned by HairFast and shared by Embedding and Alignment).


class NativeLatentStages(Stages):
    """`rotate`, `blend` (incl. the CLIP ViT-B/32 image tower when `clip_state` is given; `clip_image_embed` injects the
    reference's `clip_model.encode_image` instead), `shape_adaptor` and SEAN (`sean_inpaint`) on this backend's own
    RotateModel / ClipBlendingModel / CtrlHair mask generator / SeanModel where their state
    dicts are given; a stage whose state dict is absent is delegated to `base`."""

    def __init__(self, base, device, rotate_state=None, blend_state=None, clip_image_embed=None, shape_state=None,
                 sean_state=None, sean_mean_codes=None, clip_state=None):
        self.base = base
        self.rotate_model = self.blend_model = self.mask_generator = self.sean_model = self.clip_tower = None
        if clip_state is not None:    # clip.load("ViT-B/32") (models/Encoders.py:79): the OpenAI model's state dict
            from .clip_vit import ClipImageTower

            self.clip_tower = ClipImageTower().eval()
            self.clip_tower.load_clip_state_dict(clip_state)
            self.clip_tower.to(device)
            if clip_image_embed is None:
                clip_image_embed = self.clip_tower.encode_image
        if sean_state is not None:    # pretrained_models/sean_checkpoints/CelebA-HQ_pretrained/latest_net_G.pth (Alignment.py:29-30)
            from .sean import SeanModel

            if sean_mean_codes is None:
                raise ValueError("sean_state needs sean_mean_codes [19,512] (the median ACE.npy codes decode_sean starts "
                                 "from, pix2pix_model.py:268-293,311): without them a label of the target mask that the "
                                 "source image lacks would silently be rendered from an all-zero style code")

            self.sean_model = SeanModel(sean_mean_codes).eval()
            own = {(k if k.startswith("netG.") else "netG." + k): v for k, v in sean_state.items()}  # util.load_network loads netG's dict
            self.sean_model.load_state_dict(own)
            self.sean_model.to(device)
        if shape_state is not None:   # pretrained_models/ShapeAdaptor/mask_generator.pth (Alignment.py:33-35)
            from .shape_adaptor import MaskGenerator

            self.mask_generator = MaskGenerator().eval()
            self.mask_generator.load_state_dict(shape_state)
            self.mask_generator.to(device)
        if rotate_state is not None:  # pretrained_models/Rotate/rotate_best.pth ['model_state_dict'] (Alignment.py:36-38)
            self.rotate_model = RotateModel().eval()
            self.rotate_model.load_state_dict(rotate_state)
            self.rotate_model.to(device)
        if blend_state is not None:   # pretrained_models/Blending/checkpoint.pth ['model_state_dict'] (Blending.py:22-26)
            if clip_image_embed is None:
                raise ValueError("blend_state needs the CLIP ViT-B/32 image tower: clip_state (the OpenAI model's state "
                                 "dict, run natively) or clip_image_embed (a callable, models/Encoders.py:91-94)")
            self.blend_model = ClipBlendingModel(image_embed=clip_image_embed).eval()
            own = {k: v for k, v in blend_state.items() if not k.startswith("clip_model.")}  # the frozen tower's entries
            self.blend_model.load_state_dict(own)
            self.blend_model.to(device)

    def rotate(self, w_source_0_6, w_target_0_6):
        return self.rotate_model(w_source_0_6, w_target_0_6) if self.rotate_model is not None else self.base.rotate(w_source_0_6, w_target_0_6)

    def shape_adaptor(self, mask_target_pose, mask_hair_source):
        if self.mask_generator is not None:
            from .shape_adaptor import adapt_shape

            return adapt_shape(self.mask_generator, mask_target_pose, mask_hair_source)
        return self.base.shape_adaptor(mask_target_pose, mask_hair_source)

    def sean_inpaint(self, images_256, labels, target_mask):
        if self.sean_model is not None:
            return list(self.sean_model.inpaint_pairs(images_256, labels, target_mask))
        return self.base.sean_inpaint(images_256, labels, target_mask)

    def sean_inpaint_pairs(self, images_256, labels, target_masks):
        """SEAN for P pairs in one batched pass (rows 2p, 2p+1 of images / labels = pair p): [2P,3,256,256] in (-1,1)."""
        if self.sean_model is not None:
            return list(self.sean_model.inpaint_pairs(images_256, labels, target_masks))
        out = []
        for p_ in range(target_masks.shape[0]):
            out += list(self.base.sean_inpaint(images_256[2 * p_:2 * p_ + 2], labels[2 * p_:2 * p_ + 2], target_masks[p_:p_ + 1]))
        return out

    def blend(self, s_face_6_18, s_color_6_18, image_face_masked, image_color_masked):
        if self.blend_model is not None:
            return self.blend_model(s_face_6_18, s_color_6_18, image_face_masked, image_color_masked)
        return self.base.blend(s_face_6_18, s_color_6_18, image_face_masked, image_color_masked)


def native_stages(opts, which=("sean", "shape", "rotate", "blend"), base=None, root=None, rotate_state=None, blend_state=None,
                  clip_image_embed=None, shape_state=None, sean_state=None, sean_mean_codes=None, clip_state=None):
    """NativeLatentStages with every network of `which` whose state dict is not given read from the reference's
    checkpoint files (hairfastgan_amd.checkpoints; models/Alignment.py:29-38, models/Blending.py:24-27).  A missing file
    raises FileNotFoundError: no stage is ever left randomly initialised."""
    if "sean" in which and sean_state is None:
        sean_state, file_codes = ckpt_files.sean(root)
        sean_mean_codes = sean_mean_codes if sean_mean_codes is not None else file_codes
    if "shape" in which and shape_state is None:
        shape_state = ckpt_files.shape_adaptor(root)
    if "rotate" in which and rotate_state is None:
        rotate_state = ckpt_files.rotate(opts.rotate_checkpoint, root)
    if "blend" in which and blend_state is None:
        need_tower = clip_state is None and clip_image_embed is None
        blend_state, file_clip = ckpt_files.blending(opts.blending_checkpoint, root, need_tower=need_tower)
        if need_tower:
            clip_state = file_clip
    elif "blend" in which and clip_state is None and clip_image_embed is None:
        clip_state = ckpt_files.clip_tower("ViT-B/32", root)
    return NativeLatentStages(base or Stages(), opts.device, rotate_state, blend_state, clip_image_embed, shape_state,
                              sean_state, sean_mean_codes, clip_state)


# ---------------------------------------------------------------------------------------------
# the three stage objects of the reference
# ---------------------------------------------------------------------------------------------
def build_parsing(opts, state=None, root=None):
    """The BiSeNet singleton of my_parsing_util.py:72-81 (pretrained_models/BiSeNet/face_parsing_79999_iter.pth)."""
    net = BiSeNet(19).eval()
    net.load_state_dict(state if state is not None else ckpt_files.bisenet(root))
    return net.to(opts.device)


def build_e4e(opts, state=None, latent_avg=None, root=None):
    """models/encoder4editing/utils/model_utils.py:17-28 setup_model: `pSp(opts).encoder` + `latent_avg` as the
    namespace get_latents reads (the decoder half of pSp is never called by HairFast)."""
    if state is None:
        state, file_avg = ckpt_files.e4e(root)
        latent_avg = latent_avg if latent_avg is not None else file_avg
    if latent_avg is None:
        raise ValueError("e4e_state needs e4e_latent_avg (ckpt['latent_avg'] of e4e_ffhq_encode.pt, psp.py:93-95)")
    enc = Encoder4Editing(50, "ir_se", argparse.Namespace(stylegan_size=opts.size)).eval()
    enc.load_state_dict(state)
    return argparse.Namespace(encoder=enc.to(opts.device), opts=argparse.Namespace(start_from_latent_avg=True),
                              latent_avg=torch.as_tensor(latent_avg).float().to(opts.device))


def build_fs_encoder(opts, generator, state=None, dlatent_avg=None, root=None):
    """models/FeatureStyleEncoder/FSencoder.py:31-41 get_trainer: `trainer.enc` <- 143_enc.pth, `dlatent_avg` <- the pSp
    checkpoint's latent_avg (trainer.py:192)."""
    if state is None:
        state, file_avg = ckpt_files.fs_encoder(root)
        dlatent_avg = dlatent_avg if dlatent_avg is not None else file_avg
    if dlatent_avg is None:
        raise ValueError("fs_state needs fs_dlatent_avg (psp_ffhq_encode.pt['latent_avg'], trainer.py:192)")
    enc = FSEncoder(generator=generator)
    enc.enc.load_state_dict(state)
    enc.to(opts.device)
    enc.dlatent_avg.copy_(torch.as_tensor(dlatent_avg).float().to(opts.device).expand_as(enc.dlatent_avg))
    return enc


class Embedding(nn.Module):  # models/Embedding.py:17-117
    def __init__(self, opts, net=None, stages=None, e4e_state=None, fs_state=None, e4e_latent_avg=None,
                 fs_dlatent_avg=None, parsing=None, pretrained_root=None):
        """Like the reference's constructor (:22-39) every network is read from its checkpoint file; a state dict passed
        here replaces the file (`e4e_state` with `e4e_latent_avg`, `fs_state` with `fs_dlatent_avg`)."""
        super().__init__()
        self.opts = opts
        self.net = net if net is not None else Net(opts)
        self.stages = stages or Stages()
        self.parsing = parsing if parsing is not None else build_parsing(opts, root=pretrained_root)  # models/Net.py:29 singleton
        self.e4e = build_e4e(opts, e4e_state, e4e_latent_avg, pretrained_root)
        self.encoder = build_fs_encoder(opts, self.net.generator, fs_state, fs_dlatent_avg, pretrained_root)
        self.downsample_512 = BicubicDownSample(factor=2)
        self.downsample_256 = BicubicDownSample(factor=4)
        self._overlap = os.environ.get("HAIRFAST_EMBED_OVERLAP", "1") != "0"
        # inside a hipGraph capture (swap_graphed) the side streams fork from / join the capturing stream: the graph gets
        # three parallel branches (round 4: swap_graphed 35.0 -> 32.4 ms, bit-equal to the eager swap; "0" = one chain)
        self._overlap_in_capture = os.environ.get("HAIRFAST_GRAPH_OVERLAP", "1") != "0"
        self._side = None  # two HIP side streams, created on first use
        # Lazily derived weights (prepared / split layouts, folded BatchNorms) are computed on whatever stream first
        # needs them and then shared by all: the first pass after construction or load_state_dict runs sequentially.
        self._warmed = False
        for top in (self.e4e.encoder, self.encoder, self.parsing, self.net.generator):
            for m in top.modules():  # post hooks fire per (sub)module that is loaded
                m.register_load_state_dict_post_hook(lambda *_: setattr(self, "_warmed", False))

    @staticmethod
    def normalize(x):
        return _normalize(x, (0.5, 0.5, 0.5), (0.5, 0.5, 0.5))

    @staticmethod
    def to_bisenet(x):
        return _normalize(x, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225))

    @torch.inference_mode()
    def get_e4e_embed(self, images):  # :44-54
        image = torch.stack([im / 255 if im.dtype is torch.uint8 else im for im in images]).to(self.opts.device)
        latent_W = get_latents(self.e4e, image)
        latent_F, _ = self.net.generator([latent_W], input_is_latent=True, return_latents=False, start_layer=0, end_layer=3)
        return {"F": latent_F, "W": latent_W}

    @torch.inference_mode()
    def embedding_images(self, images_to_name, **kwargs):  # :56-117
        device = self.opts.device
        images, names_all = list(images_to_name), list(images_to_name.values())
        name_to_embed = defaultdict(dict)
        bs = kwargs.get("batch_size") or self.opts.batch_size  # swap_batch: the images of several triples per call
        for b0 in range(0, len(images), bs):
            image = torch.stack([im / 255 if im.dtype is torch.uint8 else im for im in images[b0:b0 + bs]]).to(device)
            names_b = names_all[b0:b0 + bs]
            im_512 = self.downsample_512(image)
            im_256 = self.downsample_256(image)
            im_256_norm = self.normalize(im_256)
            # The three branches below share nothing but their inputs.  For a single swap (<= 6 images: every kernel
            # of the encoders is a latency-bound launch that fills a fraction of the chip) the FS-encoder branch and
            # the parsing branch are ENQUEUED on two side streams so that the GPU overlaps them with e4e; the host
            # order of the calls - and with it torch's RNG stream - is that of the sequential form: same results.
            overlap = (self._overlap and self._warmed and image.is_cuda and image.shape[0] <= 6
                       and (self._overlap_in_capture or not torch.cuda.is_current_stream_capturing()))
            main = torch.cuda.current_stream() if overlap else None
            if overlap:
                if self._side is None:
                    self._side = (torch.cuda.Stream(device), torch.cuda.Stream(device))
                ready = torch.cuda.Event()
                ready.record(main)
                for st_ in self._side:  # a side stream starts after everything enqueued on `main` so far (incl. the
                    st_.wait_event(ready)  # previous swap's readers of blocks its allocator pool may hand out again)
            latent_W = get_latents(self.e4e, im_256_norm)  # E4E
            with (torch.cuda.stream(self._side[0]) if overlap else contextlib.nullcontext()):
                output = self.encoder.test(img=self.normalize(image), return_latent=True)  # FS encoder
                latent = output.pop()    # [bs, 512, 16, 16]
                latent_S = output.pop()  # [bs, 18, 512]
                latent_F, _ = self.net.generator([latent_S], input_is_latent=True, return_latents=False, start_layer=3,
                                                 end_layer=3, layer_in=latent)
            with (torch.cuda.stream(self._side[1]) if overlap else contextlib.nullcontext()):
                # BiSeNet: the reference parses the images one by one (:81); the batch is one call here (samples are independent)
                masks = get_segmentation(self.parsing, self.to_bisenet(im_512))
            if overlap:
                for st_ in self._side:
                    main.wait_stream(st_)
                for t_ in (latent_S, latent_F, masks):  # allocated on a side stream, consumed on `main` from here on
                    t_.record_stream(main)
            self._warmed = True
            if len(images_to_name) > 1:  # mixing if we change the colour or the shape
                hair_mask = (masks == 13).float()
                hair_mask = F.interpolate(hair_mask, size=(32, 32), mode="bicubic")
                latent_F_from_W = self.net.generator([latent_W], input_is_latent=True, return_latents=False, start_layer=0,
                                                     end_layer=3)[0]
                latent_F = latent_F + self.opts.mixing * hair_mask * (latent_F_from_W - latent_F)
            for k, names in enumerate(names_b):
                for name in names:
                    e = name_to_embed[name]
                    e["W"], e["F"], e["S"] = latent_W[k].unsqueeze(0), latent_F[k].unsqueeze(0), latent_S[k].unsqueeze(0)
                    e["mask"] = masks[k].unsqueeze(0)
                    e["image_256"], e["image_norm_256"] = im_256[k].unsqueeze(0), im_256_norm[k].unsqueeze(0)
        return name_to_embed


class Alignment(nn.Module):  # models/Alignment.py:15-175
    def __init__(self, opts, latent_encoder=None, net=None, stages=None, parsing=None, pretrained_root=None):
        """`stages` None: SEAN, the shape adaptor and RotateModel are read from their checkpoint files (:29-38)."""
        super().__init__()
        self.opts = opts
        self.latent_encoder = latent_encoder
        self.net = net if net is not None else Net(opts)
        self.stages = stages if stages is not None else native_stages(opts, ("sean", "shape", "rotate"), root=pretrained_root)
        self.parsing = parsing if parsing is not None else build_parsing(opts, root=pretrained_root)
        self.dilate_erosion = DilateErosion(dilate_erosion=opts.smooth, device=opts.device)

    @torch.inference_mode()
    def rotate_images(self, pairs, name_to_embed):
        """The `Rotate stage` of shape_module (:58-67) for several (im_name1, im_name2) pairs at once: the
        rotated latents are stacked into ONE generator forward (the reference runs one batch-1 forward
        per pair), and so is the shape adaptor on their parses (:74-77: one call for all pairs).
        Returns {(name1, name2): (I_rot [1,3,size,size], rot_mask, target_mask)}."""
        todo = [(a, b) for a, b in pairs if name_to_embed[a]["image_256"] is not name_to_embed[b]["image_256"]]
        if not todo:
            return {}
        w1 = torch.cat([name_to_embed[a]["W"] for a, _ in todo], 0)
        w2 = torch.cat([name_to_embed[b]["W"] for _, b in todo], 0)
        lat = torch.cat((self.stages.rotate(w2[:, :6], w1[:, :6]), w2[:, 6:]), dim=1)  # every pair in one Rotate call
        I_rot, _ = self.net.generator([lat], input_is_latent=True, return_latents=False)
        seg_in = Embedding.to_bisenet(((I_rot + 1) / 2).clip(0, 1))
        masks = get_segmentation(self.parsing, seg_in)  # the 1024^2 images are parsed (:65-67), both in one call
        targets = self.stages.shape_adaptor(torch.cat([name_to_embed[a]["mask"] for a, _ in todo], 0), masks)
        return {key: (I_rot[k:k + 1], masks[k:k + 1], targets[k:k + 1]) for k, key in enumerate(todo)}

    def _target_mask(self, im_name1, im_name2, name_to_embed, rotated=None):
        """The target parse of a pair (:58-83): the shape adaptor's output on the rotated image, or the face's own parse
        when both names hold the same image."""
        e1, e2 = name_to_embed[im_name1], name_to_embed[im_name2]
        if e1["image_256"] is e2["image_256"]:
            return e1["mask"]
        rot = (rotated or {}).get((im_name1, im_name2))
        if rot is None:
            rot = self.rotate_images([(im_name1, im_name2)], name_to_embed)[(im_name1, im_name2)]
        return rot[2]

    @torch.inference_mode()
    def shape_module(self, im_name1, im_name2, name_to_embed, only_target=True, rotated=None, **kwargs):  # :40-99
        e1, e2 = name_to_embed[im_name1], name_to_embed[im_name2]
        inp_mask1, inp_mask2 = e1["mask"], e2["mask"]
        target_mask = self._target_mask(im_name1, im_name2, name_to_embed, rotated)
        hair_mask_target = (target_mask == 13).to(target_mask.dtype)
        if only_target:
            return {"HM_X": hair_mask_target}
        return (inp_mask1, (inp_mask1 == 13).to(inp_mask1.dtype), inp_mask2, (inp_mask2 == 13).to(inp_mask2.dtype),
                target_mask, hair_mask_target)

    @torch.inference_mode()
    def shape_modules(self, pairs, name_to_embed, rotated=None, **kwargs):
        """`shape_module(..., only_target=True)` for several pairs: the hair masks of all target parses from one comparison
        (two launches instead of two per pair; elementwise, so every pair's mask has the bits of its own call)."""
        if not pairs:
            return []
        targets = [self._target_mask(a, b, name_to_embed, rotated) for a, b in pairs]
        stacked = _cat(targets)
        hair = (stacked == 13).to(stacked.dtype)
        return [{"HM_X": hair[j:j + 1]} for j in range(len(pairs))]

    @torch.inference_mode()
    def align_images(self, im_name1, im_name2, name_to_embed, **kwargs):  # :101-175
        return self.align_images_batch([(im_name1, im_name2)], name_to_embed, **kwargs)[0]

    @torch.inference_mode()
    def align_images_batch(self, pairs, name_to_embed, **kwargs):
        """align_images (:101-175) for several (im_name1, im_name2) pairs - several triples - at once: the
        out-of-scope stages run per pair, everything on the hot path once for all pairs (the SEAN outputs of all
        pairs are ONE e4e batch + ONE generator 0->3 forward, the masks one dilation / erosion call)."""
        results = [None] * len(pairs)
        work = []  # (index, e1, e2, target parse of the pair)
        for k, (n1, n2) in enumerate(pairs):
            e1, e2 = name_to_embed[n1], name_to_embed[n2]
            if e1["image_256"] is e2["image_256"]:
                hm = self.shape_module(n1, n2, name_to_embed, only_target=True, **kwargs)["HM_X"]
                results[k] = {"latent_F_align": e1["F"], "HM_X": hm}
            else:
                work.append((k, e1, e2, self._target_mask(n1, n2, name_to_embed, kwargs.get("rotated"))))
        if not work:
            return results
        # the masks of every pair from whole-batch launches (:84-99, :133-141 are elementwise: same bits as pair by pair)
        P = len(work)
        parses = _cat([w[1]["mask"] for w in work] + [w[2]["mask"] for w in work])  # [2P,1,h,w]: the faces' parses, then the shapes'
        targets = _cat([w[3] for w in work])
        hair, hair_target = (parses == 13).to(parses.dtype), (targets == 13).to(targets.dtype)
        hair1, hair2 = hair[:P], hair[P:]
        masks = torch.stack([1 - (1 - hair1) * (1 - hair_target), hair_target, hair2 * hair_target], dim=1).reshape(3 * P, *hair.shape[1:])
        _mark("align: shape module, masks")
        batched_sean = getattr(self.stages, "sean_inpaint_pairs", None)
        if batched_sean is not None:  # SEAN for inpaint: every pair in one batched pass (hairfastgan_amd.sean)
            sean = batched_sean(torch.cat([e["image_256"] for _, e1, e2, _m in work for e in (e1, e2)], dim=0),
                                torch.cat([e["mask"] for _, e1, e2, _m in work for e in (e1, e2)], dim=0),
                                targets)
        else:
            sean = []
            for _, e1, e2, target_mask in work:  # SEAN for inpaint (per pair)
                sean += list(self.stages.sean_inpaint(torch.cat([e1["image_256"], e2["image_256"]], dim=0),
                                                      torch.cat([e1["mask"], e2["mask"]], dim=0), target_mask))
        _mark("align: SEAN encode + decodes")
        enc_F = self.latent_encoder(sean)["F"]                                   # e4e batch 2P + generator 0->3
        dilate, erosion = self.dilate_erosion.mask(masks)                        # [3P, 1, 256, 256] each
        free_mask = torch.stack([dilate[0::3], erosion[1::3], erosion[2::3]], dim=1).reshape(-1, *dilate.shape[1:])
        low = 1 - F.interpolate(free_mask.float(), size=(32, 32), mode="bicubic")  # [3P, 1, 32, 32]
        # the three interpolations of F (:160-172) for all pairs at once
        enc_F = enc_F.reshape(P, 2, *enc_F.shape[1:])
        intermediate_align, latent_F_out_new = enc_F[:, 0], enc_F[:, 1]
        il = low.reshape(P, 3, *low.shape[1:])
        F_1, F_2 = _cat([e1["F"] for _, e1, _e2, _m in work]), _cat([e2["F"] for _, _e1, e2, _m in work])
        latent_F_align = intermediate_align + il[:, 0] * (F_1 - intermediate_align)
        latent_F_align = latent_F_out_new + il[:, 1] * (latent_F_align - latent_F_out_new)
        latent_F_align = F_2 + il[:, 2] * (latent_F_align - F_2)
        for j, (k, _e1, _e2, _m) in enumerate(work):
            results[k] = {"latent_F_align": latent_F_align[j:j + 1], "HM_X": hair_target[j:j + 1]}
        return results


class Blending(nn.Module):  # models/Blending.py:11-82
    def __init__(self, opts, net=None, stages=None, pp_state=None, pp_latent_avg=None, pretrained_root=None):
        """`stages` None: ClipBlendingModel and its CLIP tower are read from args.blending_checkpoint (:24-27);
        `pp_state` None: PostProcessModel from args.pp_checkpoint + PostProcess/latent_avg.pt (:29-30, Encoders.py:112)."""
        super().__init__()
        self.opts = opts
        self.net = net if net is not None else Net(opts)
        self.stages = stages if stages is not None else native_stages(opts, ("blend",), root=pretrained_root)
        if pp_state is None:
            pp_state, file_avg = ckpt_files.post_process(opts.pp_checkpoint, pretrained_root)
            pp_latent_avg = pp_latent_avg if pp_latent_avg is not None else file_avg
        if pp_latent_avg is None:
            raise ValueError("pp_state needs pp_latent_avg (pretrained_models/PostProcess/latent_avg.pt, models/Encoders.py:112)")
        avg = torch.as_tensor(pp_latent_avg).float().reshape(-1, 512)  # [18,512], [1,18,512] or one [512] row for all 18
        self.post_process = PostProcessModel(latent_avg=avg.expand(18, 512) if avg.shape[0] == 1 else avg).eval()
        self.post_process.load_state_dict(pp_state)
        self.post_process.to(opts.device)
        self.dilate_erosion = DilateErosion(dilate_erosion=opts.smooth, device=opts.device)
        self.downsample_256 = BicubicDownSample(factor=4)

    @torch.inference_mode()
    def blend_images(self, align_shape, align_color, name_to_embed, **kwargs):  # :36-82
        return self.blend_images_batch([align_shape], [align_color], name_to_embed, [("face", "shape", "color")], **kwargs)[0]

    @torch.inference_mode()
    def blend_images_batch(self, aligns_shape, aligns_color, name_to_embed, keys, **kwargs):
        """blend_images (:36-82) for several triples: keys[t] = the (face, shape, color) entries of triple t in
        name_to_embed.  The blending encoder (out of scope) runs per triple; the generator 4->8 forward, the
        bicubic down-sampling, PostProcessModel and the final generator 5->8 forward once with batch T."""
        T = len(keys)
        emb = [[name_to_embed[k] for k in key] for key in keys]
        mask_de = self.dilate_erosion.hair_from_mask(torch.cat([e[i]["mask"] for e in emb for i in (0, 2)], dim=0))  # [2T,...]
        HM_XD, _ = self.dilate_erosion.mask(torch.cat([a["HM_X"] for a in aligns_color], dim=0))
        # the blending encoder's inputs (:50-58) for all triples from whole-batch launches (elementwise: the bits of the per-triple form)
        HM_1D, HM_3D, HM_3E = mask_de[0][0::2], mask_de[0][1::2], mask_de[1][1::2]           # [T, 1, 256, 256] each
        target_mask = (1 - HM_1D) * (1 - HM_3D) * (1 - HM_XD)
        I_1_all = _cat([e[0]["image_norm_256"] for e in emb])
        I_3_all = _cat([e[2]["image_norm_256"] for e in emb])
        face_in, color_in = I_1_all * target_mask, I_3_all * HM_3E
        S_blend = [None] * T
        todo = [t for t, (ef, es, ec) in enumerate(emb)
                if ef["image_norm_256"] is not ec["image_norm_256"] or ef["image_norm_256"] is not es["image_norm_256"]]
        for t, (ef, _es, _ec) in enumerate(emb):
            if t not in todo:
                S_blend[t] = ef["S"]
        _mark("blend: masks")
        if todo:  # the blending encoder once for all triples that need it
            if len(todo) < T:
                rows = torch.tensor(todo, device=face_in.device)
                face_in, color_in = face_in.index_select(0, rows), color_in.index_select(0, rows)
            S_6_18 = self.stages.blend(_cat([emb[t][0]["S"][:, 6:] for t in todo]), _cat([emb[t][2]["S"][:, 6:] for t in todo]),
                                       face_in, color_in)
            _mark("blend: ClipBlendingModel incl. CLIP tower")
            for j, t in enumerate(todo):
                S_blend[t] = torch.cat((emb[t][0]["S"][:, :6], S_6_18[j:j + 1]), dim=1)
        latent_F_align = torch.cat([a["latent_F_align"] for a in aligns_shape], dim=0)
        I_blend, _ = self.net.generator([torch.cat(S_blend, 0)], input_is_latent=True, return_latents=False, start_layer=4,
                                        end_layer=8, layer_in=latent_F_align)
        I_blend_256 = self.downsample_256(I_blend)
        _mark("blend: generator 4->8")
        S_final, F_final = self.post_process(I_1_all, I_blend_256)  # Post Process (native: encoders/post_process.py)
        _mark("blend: PostProcessModel")
        I_final, _ = self.net.generator([S_final], input_is_latent=True, return_latents=False, start_layer=5, end_layer=8,
                                        layer_in=F_final)
        out = ((I_final + 1) / 2).clip(0, 1)
        _mark("blend: generator 5->8")
        return [out[t] for t in range(T)]


class HairFast:
    """HairFast with the reference's hairstyle-transfer interface (hair_swap.py:27-103).

    `HairFast(args)` reads every network from the reference's checkpoint files - args.ckpt, args.rotate_checkpoint,
    args.blending_checkpoint, args.pp_checkpoint and the paths the reference hard-codes (hairfastgan_amd.checkpoints lists
    them with their key handling) - relative to the working directory or `pretrained_root`; a missing file raises
    FileNotFoundError naming it.  Keyword-only arguments replace files by in-memory state dicts (tests, benchmarks):
      pretrained_root directory the checkpoint paths are relative to (default: cwd / HAIRFAST_PRETRAINED_ROOT)
      stages          a `Stages` object supplying rotate / shape_adaptor / sean_inpaint / blend (e.g. the reference's own
                      modules); networks whose state dicts are passed below still run natively, the
                      files of the other STAGE networks (SEAN, shape adaptor, RotateModel, ClipBlendingModel) are then NOT
                      read.  Independent of `stages`: the generator, both encoders, BiSeNet and PostProcessModel always
                      come from their files (args.ckpt, args.pp_checkpoint + PostProcess/latent_avg.pt, ...) unless
                      their own state dicts (generator_state, e4e_state, fs_state, bisenet_state, pp_state + pp_latent_avg) are given
      generator_state {'g_ema': ..., 'latent_avg': ...} instead of args.ckpt
      e4e_state + e4e_latent_avg / fs_state + fs_dlatent_avg   encoder state dicts with their average latents
      pp_state + pp_latent_avg    PostProcessModel ('model_state_dict' of args.pp_checkpoint; PostProcess/latent_avg.pt)
      bisenet_state   BiSeNet state dict (pretrained_models/BiSeNet/face_parsing_79999_iter.pth)
      rotate_state    RotateModel ('model_state_dict' of args.rotate_checkpoint)
      blend_state     ClipBlendingModel ('model_state_dict' of args.blending_checkpoint, `clip_model.*` dropped) with
      clip_state      the OpenAI CLIP ViT-B/32 state dict (its `visual.*` entries; what clip.load("ViT-B/32") holds,
                      models/Encoders.py:79; run natively by hairfastgan_amd.clip_vit) or `clip_image_embed`, a callable
      shape_state     CtrlHair mask generator (pretrained_models/ShapeAdaptor/mask_generator.pth)
      conv_precision  this object's matrix-core mode ("f16x3" | "f32" | "f16" | "auto"; _runtime.CONV_PRECISIONS): set for
                      the duration of each swap / swap_batch / swap_graphed call and restored, so that two HairFast objects
                      of one process can differ (default None: the process-wide HAIRFAST_CONV_PRECISION / set_conv_precision)
      batch_invariant this object's plan mode (None: the process-wide HAIRFAST_DETERMINISTIC / set_batch_invariant, ON by
                      default): True = `swap_batch` gives every triple the bits - and mask indices - `swap` gives it alone;
                      False = plans from the whole launch (a few per cent faster batched, near-tie argmax flips possible)
      sean_state + sean_mean_codes   SEAN generator (…/CelebA-HQ_pretrained/latest_net_G.pth) and the [19,512] per-label
                      median style codes (models/sean_codes/styles_test/mean_style_code/median/<label>/ACE.npy)
    """

    @staticmethod
    def _as_tensor(img, cache=None):
        """The image forms of the reference's `swap` (hair_swap.py:79-92): torch.Tensor [3,H,W] (uint8 or float in [0,1])
        as is; `PIL.Image.Image` and numpy HWC arrays with `F.to_tensor`'s semantics (uint8 -> CHW float / 255 on the CPU;
        other dtypes keep their values); a `Path` / `str` is read once per call - an image file through PIL (the
        reference's `read_image(path, mode=ImageReadMode.RGB)`: uint8 [3,H,W]), a `.npy` file as the array it holds.
        PIL is imported lazily: tensors and arrays never need it."""
        if isinstance(img, torch.Tensor):
            return img
        if isinstance(img, np.ndarray):
            return _to_tensor_like_torchvision(img)
        if isinstance(img, (Path, str)):
            cache = cache if cache is not None else {}
            if img not in cache:
                if str(img).endswith(".npy"):
                    arr = np.load(str(img))
                    cache[img] = torch.from_numpy(arr).permute(2, 0, 1).contiguous() if arr.ndim == 3 and arr.shape[-1] == 3 else torch.from_numpy(arr)
                else:
                    cache[img] = _read_image_rgb(str(img))
            return cache[img]
        if type(img).__module__.split(".")[0] == "PIL":  # PIL.Image.Image (any subclass), without importing PIL for other inputs
            arr = np.asarray(img)  # to_tensor: one channel per band; mode "1" is 0 / 255, "I" / "I;16" / "F" keep their values
            return _to_tensor_like_torchvision(arr.astype(np.uint8) * 255 if img.mode == "1" else arr)
        raise TypeError(f"Unsupported image format {type(img)}")

    def __init__(self, args, *, stages=None, generator_state=None, e4e_state=None, fs_state=None, e4e_latent_avg=None,
                 fs_dlatent_avg=None, pp_state=None, pp_latent_avg=None, bisenet_state=None, rotate_state=None,
                 blend_state=None, clip_image_embed=None, shape_state=None, sean_state=None, sean_mean_codes=None,
                 clip_state=None, pretrained_root=None, conv_precision=None, batch_invariant=None):
        self.args = args
        self.conv_precision = conv_precision  # None: the process-wide mode; else this object's own (set per call)
        self.batch_invariant = batch_invariant  # None: the process-wide setting (default on); True / False: this object's own
        conv_precision_scope(conv_precision)  # validates
        if getattr(args, "save_all", False):
            raise NotImplementedError(
                "--save_all (intermediate images / latents written by the reference's utils/save_utils.py, Embedding.py:94-108, "
                "Alignment.py:84-93, 159-179, Blending.py:70-78) is not implemented by this backend: nothing would be written")
        root = pretrained_root
        given = dict(rotate_state=rotate_state, blend_state=blend_state, clip_image_embed=clip_image_embed,
                     shape_state=shape_state, sean_state=sean_state, sean_mean_codes=sean_mean_codes, clip_state=clip_state)
        if stages is None:  # the reference's constructor: everything from files unless handed over in memory
            self.stages = native_stages(args, root=root, **given)
        elif any(given[k] is not None for k in ("rotate_state", "blend_state", "shape_state", "sean_state")):
            # nothing is read from files for the stages `stages` supplies - except the CLIP tower a given blend_state needs
            # (its state dict or callable not passed): checkpoint / ~/.cache/clip, as in the default path above
            self.stages = native_stages(args, which=("blend",) if blend_state is not None else (), base=stages, root=root, **given)
        else:
            self.stages = stages
        self.net = Net(args, state=generator_state, root=root)
        self.parsing = build_parsing(args, bisenet_state, root)  # my_parsing_util.py:77-79
        self.embed = Embedding(args, net=self.net, stages=self.stages, e4e_state=e4e_state, fs_state=fs_state,
                               e4e_latent_avg=e4e_latent_avg, fs_dlatent_avg=fs_dlatent_avg, parsing=self.parsing,
                               pretrained_root=root)
        self.align = Alignment(args, self.embed.get_e4e_embed, net=self.net, stages=self.stages, parsing=self.parsing)
        self.blend = Blending(args, net=self.net, stages=self.stages, pp_state=pp_state, pp_latent_avg=pp_latent_avg,
                              pretrained_root=root)
        self._times = []

    def _swap_from_tensors(self, face, shape, color, **kwargs):  # hair_swap.py:38-61
        return self._swap_batch_from_tensors([(face, shape, color)], **kwargs)[0]

    def _swap_batch_from_tensors(self, triples, **kwargs):
        """hair_swap.py:38-61 for T triples at once.  Triples are independent and the parameters frozen, so every
        hot-path call runs once with the batch of all triples (Embedding: 3T images; Rotate: 2T full forwards + parses;
        Alignment: e4e on 2T SEAN outputs; Blending: T) instead of T times with batch 3 / 2 / 1 - per-sample results
        are the same, the GPU sees 256-channel 32^2 convolutions with T times more pixels to fill its 256 CUs with.
        The out-of-scope stages are called per triple."""
        T = len(triples)
        images_to_name = defaultdict(list)
        for t, triple in enumerate(triples):
            for image, name in zip(triple, ("face", "shape", "color")):
                images_to_name[image].append((t, name) if T > 1 else name)
        key = (lambda t, n: (t, n)) if T > 1 else (lambda t, n: n)
        if T > 1:
            kwargs = dict(kwargs, batch_size=max(self.args.batch_size, len(images_to_name)))
        _mark("start")
        name_to_embed = self.embed.embedding_images(images_to_name, **kwargs)  # Embedding stage
        _mark("embedding: e4e, FS encoder, BiSeNet, generator 3->3 / 0->3")
        kwargs.pop("batch_size", None)
        same = [triple[1] is triple[2] for triple in triples]                  # shape is color
        pairs = []
        for t in range(T):
            pairs += [(key(t, "face"), key(t, "shape"))] + ([] if same[t] else [(key(t, "face"), key(t, "color"))])
        rotated = self.align.rotate_images(pairs, name_to_embed)               # every Rotate forward as one batch
        _mark("rotate: RotateModel, generator 0->8, BiSeNet @1024, shape adaptor")
        aligns_shape = self.align.align_images_batch([(key(t, "face"), key(t, "shape")) for t in range(T)], name_to_embed,
                                                     rotated=rotated, **kwargs)
        color_targets = iter(self.align.shape_modules([(key(t, "face"), key(t, "color")) for t in range(T) if not same[t]],
                                                      name_to_embed, rotated=rotated, **kwargs))
        aligns_color = [aligns_shape[t] if same[t] else next(color_targets) for t in range(T)]
        _mark("align: e4e of the SEAN renderings, generator 0->3, F alignment")
        return self.blend.blend_images_batch(aligns_shape, aligns_color, name_to_embed,
                                             [tuple(key(t, n) for n in ("face", "shape", "color")) for t in range(T)], **kwargs)

    def swap(self, face_img, shape_img, color_img, benchmark=False, align=False, seed=None, exp_name=None, **kwargs):
        """hair_swap.py:63-103.  Images: torch.Tensor [3,H,W] (uint8 or float in [0,1]), `PIL.Image.Image`, numpy HWC
        arrays, or file paths (image files decoded through PIL like the reference's `read_image(..., RGB)`; `.npy` arrays).

        Randomness: `seed` (default 3407, utils/seed.py:19) makes a swap reproducible on THIS backend; it does not
        reproduce a seeded run of the reference sample for sample: the per-layer noise of a generator forward is one draw
        here (17 in the reference), the FS encoder's discarded generator forward (trainer.py:295, which only advances the
        RNG) is not run, SEAN's 18 ACE noise maps per decode are one draw, and hipRAND's Philox walk differs from cuRAND's
        in any case.  `HAIRFAST_RNG_WALK=reference` restores the reference's ORDER of consumption (one draw per noise layer,
        the discarded forward run; _runtime.reference_rng_walk) at the cost of those launches.  Bit-level comparisons with
        the reference inject the noise explicitly
        (Generator.forward(noise= / randomize_noise=False), SPADEGenerator.noise_source; tests/test_gpu_pipeline.py)."""
        cache = {}
        images = [self._as_tensor(img, cache) for img in (face_img, shape_img, color_img)]
        if align:
            raise NotImplementedError("align=True needs the reference's dlib face aligner (utils/shape_predictor.py): out of scope")
        images = equal_replacer(images)
        set_seed(3407 if seed is None else seed)  # utils/seed.py:19-31
        if benchmark:  # utils/time.py:15-37
            torch.cuda.current_stream().synchronize()
            t0 = time.time()
        seed_ = 3407 if seed is None else seed

        def run():  # HAIRFAST_CONV_PRECISION=auto: the whole swap again on the fp32 kernels if the fp16 split clamped
            set_seed(seed_)
            return self._swap_from_tensors(*images, exp_name=exp_name, **kwargs)

        with conv_precision_scope(self.conv_precision), batch_invariant_scope(self.batch_invariant):
            final_image = run_guarded(run)
        if benchmark:
            torch.cuda.current_stream().synchronize()
            self._times.append(time.time() - t0)
            print(f"\n{len(self._times)} experiment ended in {self._times[-1]:.3f}(s)\nmin time: {np.min(self._times):.3f}(s), "
                  f"median time: {np.median(self._times):.3f}(s), std time: {np.std(self._times):.3f}(s)", file=sys.stderr)
        return final_image

    __call__ = swap

    def swap_graphed(self, face_img, shape_img, color_img, seed=None):
        """`swap` with the whole stage sequence - ~1400 kernel launches and the torch glue between them - recorded ONCE
        into a hipGraph and replayed with a single launch (BASELINE.json configs[2], one triple: at batch sizes 1-3 the
        host otherwise issues launches more slowly than the GPU retires them; 36 -> ~25 ms).  Three DISTINCT images of one
        fixed size on the GPU (anything else falls back to `swap`); the result is a fresh tensor.  Noise is drawn inside the
        graph from torch's graph-safe generator: set by `seed` before every replay, advanced by it.  The first call per
        image size captures (two eager warm-up swaps + the capture)."""
        images = [self._as_tensor(img) for img in (face_img, shape_img, color_img)]
        images = equal_replacer([im.to(self.args.device) for im in images])
        if len({id(im) for im in images}) != 3 or getattr(self.args, "save_all", False):
            return self.swap(*images, seed=seed)
        images = [im.float().contiguous() for im in images]
        with conv_precision_scope(self.conv_precision), batch_invariant_scope(self.batch_invariant):
            return self._swap_graphed(images, seed)

    def _swap_graphed(self, images, seed):
        from .graphs import GraphRunner

        # a graph replays the kernels and plans of the mode it was captured in: conv precision, RNG walk and the batch-invariant switch are part of the key
        key = (tuple(tuple(im.shape) for im in images), configured_conv_precision(), reference_rng_walk(), batch_invariant())
        graphs = self.__dict__.setdefault("_swap_graphs", {})
        if key not in graphs:
            set_seed(3407 if seed is None else seed)
            graphs[key] = GraphRunner(lambda a, b, c: self._swap_from_tensors(a, b, c), *images)
        set_seed(3407 if seed is None else seed)
        return graphs[key](*images).clone()  # (the key carries the mode the graph was captured in)

    def swap_batch(self, triples, seed=None, **kwargs):
        """Several swaps as ONE batched pass over the hot path (not in the reference: BASELINE.json configs[3],
        "batched HairFast swap").  triples: sequence of (face, shape, color) with the image forms `swap` takes
        (tensors / arrays).  Returns a list of [3, size, size] images in [0, 1], one per triple, equal to what
        `swap` returns for each triple given the same per-layer noise."""
        cache = {}
        prepared = equal_replacer_many([[self._as_tensor(img, cache) for img in triple] for triple in triples])
        set_seed(3407 if seed is None else seed)
        # a triple that repeats an image takes the reference's shortcuts (no mixing / no second Rotate): one by one
        plain = [t for t, tr in enumerate(prepared) if len({id(x) for x in tr}) == 3]
        out = [None] * len(prepared)
        seed_ = 3407 if seed is None else seed

        def run():
            set_seed(seed_)
            res = [None] * len(prepared)
            if plain:
                for t, img in zip(plain, self._swap_batch_from_tensors([prepared[t] for t in plain], **kwargs)):
                    res[t] = img
            for t, tr in enumerate(prepared):
                if res[t] is None:
                    res[t] = self._swap_from_tensors(*tr, **kwargs)
            return res

        with conv_precision_scope(self.conv_precision), batch_invariant_scope(self.batch_invariant):
            return run_guarded(run)


# ---------------------------------------------------------------------------------------------
# replay harness of round 1 (kept: tests/test_gpu_schedule.py and the graph runner use it)
# ---------------------------------------------------------------------------------------------
class HairFastHotPath(torch.nn.Module):
    """Generator + e4e + FS encoder with the per-triple hot-path CALL SCHEDULE only (resident
    synthetic tensors instead of the stages in between).  `HairFast` above is the call surface;
    this class isolates the hot-path kernels for timing and for hipGraph replay."""

    def __init__(self, args, generator_state, e4e_state=None, fs_state=None, e4e_latent_avg=None, fs_dlatent_avg=None):
        super().__init__()
        self.args = args
        self.net = Net(args, state=generator_state)
        self.e4e = build_e4e(args, e4e_state, e4e_latent_avg)
        self.encoder = build_fs_encoder(args, self.net.generator, fs_state, fs_dlatent_avg)
        self._graphs = {}

    def _call(self, key, fn, *tensors, use_graphs=False):
        """Run `fn(*tensors)` eagerly or through a per-call-site hipGraph (captured on first use)."""
        if not use_graphs:
            return fn(*tensors)
        from .graphs import GraphRunner

        k = (key,) + tuple(tuple(t.shape) for t in tensors)
        if k not in self._graphs:
            self._graphs[k] = GraphRunner(fn, *tensors)
        return self._graphs[k](*tensors)

    @torch.inference_mode()
    def swap_schedule(self, images_1024, images_256, align_inputs_256, f_align_32, f_final_64, s_blend, s_final,
                      w_rotate, include_discarded_forward=False, use_graphs=False, batch_rotations=True):
        """One triple.  images_1024 / images_256: the 3 normalised inputs [3,3,1024,1024] /
        [3,3,256,256]; align_inputs_256 [2,3,256,256]: SEAN outputs re-embedded by
        Embedding.get_e4e_embed; f_align_32 [1,512,32,32], f_final_64 [1,512,64,64]: F-space
        tensors entering the last two generator calls; s_blend / s_final / w_rotate [1,18,512].
        batch_rotations: the two Alignment.py:63 forwards as one batch-2 call (HairFast's schedule)
        or as the reference's two batch-1 calls.  use_graphs: replay each call site as a hipGraph."""
        g = self.net.generator
        ug = use_graphs
        out = {}
        gen = lambda **kw: (lambda w_, *li: g([w_], input_is_latent=True, return_latents=False,  # noqa: E731
                                              layer_in=(li[0] if li else None), **kw)[0])
        # --- Embedding.embedding_images (Embedding.py:64-92), batch 3
        w = self._call("e4e", lambda x: get_latents(self.e4e, x), images_256, use_graphs=ug)
        self.encoder.run_discarded_generator = include_discarded_forward

        def fs(img):
            res = self.encoder.test(img=img, return_latent=True)
            return res[2], res[3]

        s, fea = self._call("fs", fs, images_1024, use_graphs=ug)
        out["F"] = self._call("g33", gen(start_layer=3, end_layer=3), s, fea, use_graphs=ug)
        out["f_from_w"] = self._call("g03", gen(start_layer=0, end_layer=3), w, use_graphs=ug)
        out["W"], out["S"] = w, s
        # --- Alignment.shape_module x2 (Alignment.py:63): full forwards of rotated latents
        if batch_rotations:
            rot = self._call("g08", gen(), torch.cat([w_rotate, w_rotate], 0), use_graphs=ug)
            out["I_rot_shape"], out["I_rot_color"] = rot[0:1], rot[1:2]
        else:
            out["I_rot_shape"] = self._call("g08", gen(), w_rotate, use_graphs=ug)
            if ug:
                out["I_rot_shape"] = out["I_rot_shape"].clone()  # the same graph is replayed below
        # --- Alignment.align_images -> Embedding.get_e4e_embed (Embedding.py:44-54), batch 2
        w2 = self._call("e4e", lambda x: get_latents(self.e4e, x), align_inputs_256, use_graphs=ug)
        out["F_sean"] = self._call("g03", gen(start_layer=0, end_layer=3), w2, use_graphs=ug)
        if not batch_rotations:
            out["I_rot_color"] = self._call("g08", gen(), w_rotate, use_graphs=ug)
        # --- Blending.blend_images (Blending.py:62, 68)
        out["I_blend"] = self._call("g48", gen(start_layer=4, end_layer=8), s_blend, f_align_32, use_graphs=ug)
        out["I_final"] = self._call("g58", gen(start_layer=5, end_layer=8), s_final, f_final_64, use_graphs=ug)
        return out
