"""TEST INFRASTRUCTURE - CPU restatement of the reference's PostProcessModel.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (hairfastgan_amd/) never does.  Functional torch-CPU fp32 restatement of

  models/Encoders.py:106-137   PostProcessModel.forward
  models/Encoders.py:13-32     ModulationModule
  models/Encoders.py:35-57     FeatureiResnet([[1024, 2], [768, 2], [512, 2]])
  models/Net.py:396-477        FeatureEncoderMult(fs_layers=[9]).forward (content after block_2, stride 1)
  models/Net.py:160-190        IBasicBlock
  models/stylegan2/model.py:16-21 PixelNorm (over dim 1 of [B,18,512] - as the reference applies it)

pinned by oracle/make_golden.py against the imported reference (tests/golden/postprocess.npz,
oracle_vs_reference.json: bit-identical).  Parameters are a flat dict with the reference's
state-dict keys (`encoder_face.*`, `to_feature.res_blocks.*`, `to_latent_1.*`, `to_latent_2.*`)
plus `latent_avg` (a tensor attribute in the reference, loaded from a file).

Also the two latent-space models of SURVEY.md section 8 row f4 that are built from the same ModulationModule:
  models/Encoders.py:60-72     RotateModel.forward
  models/Encoders.py:75-103    ClipBlendingModel.forward / get_image_embed (the CLIP image tower itself is an
                               un-vendored dependency of the reference: a callable here)
pinned the same way (tests/golden/latent_models.npz).
"""
import torch
import torch.nn.functional as F

from .ref_encoders import IRESNET50_LAYERS, _bn_shapes, bn, ibasic_block

FEATURE_BLOCKS = [(1024, 2), (768, 2), (512, 2)]  # models/Encoders.py:112


def feature_encoder_mult(P, pre, x):
    """FeatureEncoderMult(fs_layers=[9]).forward on a [B,3,256,256] input (transform_to_256 is the
    identity at that size, Net.py:12-14,447): returns (latents [B,18,512], content [B,512,64,64])."""
    assert x.shape[-2:] == (256, 256)
    x = F.conv2d(x, P[f"{pre}.conv.0.weight"], padding=1)
    x = F.prelu(bn(P, f"{pre}.conv.1", x), P[f"{pre}.conv.2.weight"])
    pooled, content = [], None
    for li, (planes, nblocks) in enumerate(IRESNET50_LAYERS):
        for j in range(nblocks):
            x = ibasic_block(P, f"{pre}.block_{li + 1}.{j}", x, 2 if j == 0 else 1)
        if li == 1:  # max(fs_layers) > 7: the content branches off block_2's output (Net.py:454-460)
            cl = f"{pre}.content_layer.0"
            c = bn(P, f"{cl}.0", x)
            c = F.conv2d(c, P[f"{cl}.1.weight"], padding=1)
            c = F.prelu(bn(P, f"{cl}.2", c), P[f"{cl}.3.weight"])
            c = F.conv2d(c, P[f"{cl}.4.weight"], stride=1, padding=1)  # fs_kernals[7] = 3x3, fs_strides[7] = 1
            content = bn(P, f"{cl}.5", c)
        pooled.append(F.adaptive_avg_pool2d(x, (3, 3)))
    flat = torch.cat(pooled, dim=1).reshape(x.shape[0], -1)
    s = torch.stack([F.linear(flat, P[f"{pre}.styles.{i}.weight"], P[f"{pre}.styles.{i}.bias"]) for i in range(18)], dim=1)
    return s, content


def modulation_module(P, pre, x, embedding, layernum, last):
    """ModulationModule.forward (Encoders.py:24-32)."""
    x = F.linear(x, P[f"{pre}.fc.weight"], P[f"{pre}.fc.bias"])
    x = F.layer_norm(x, [layernum, 512])

    def mlp(name):
        h = F.linear(embedding, P[f"{pre}.{name}.0.weight"], P[f"{pre}.{name}.0.bias"])
        h = F.layer_norm(h, [h.shape[-1]], P[f"{pre}.{name}.1.weight"], P[f"{pre}.{name}.1.bias"])
        return F.linear(F.leaky_relu(h), P[f"{pre}.{name}.3.weight"], P[f"{pre}.{name}.3.bias"])

    out = x * (1 + mlp("gamma_function")) + mlp("beta_function")
    return out if last else F.leaky_relu(out)


def pixel_norm(x):
    return x * torch.rsqrt(torch.mean(x ** 2, dim=1, keepdim=True) + 1e-8)


def feature_iresnet(P, pre, x):
    """FeatureiResnet.forward (Encoders.py:54-57): six stride-1 IBasicBlocks, 1024 -> 768 -> 512 channels."""
    for n, (planes, num) in enumerate(FEATURE_BLOCKS, start=1):
        for k in range(1, num + 1):
            x = ibasic_block(P, f"{pre}.res_blocks.res_block_{n}_{k}", x, 1)
    return x


def post_process_forward(P, source, target):
    """PostProcessModel.forward (Encoders.py:119-137): (S_final [B,18,512], F_final [B,512,64,64])."""
    s_face, f_face = feature_encoder_mult(P, "encoder_face", source)
    s_hair, f_hair = feature_encoder_mult(P, "encoder_face", target)
    dt_face, dt_hair = pixel_norm(s_face), pixel_norm(s_hair)
    for i in range(5):
        dt_face = modulation_module(P, f"to_latent_1.{i}", dt_face, s_hair, 18, i == 4)
    for i in range(5):
        dt_hair = modulation_module(P, f"to_latent_2.{i}", dt_hair, s_face, 18, i == 4)
    final_s = P["latent_avg"] + 0.1 * (dt_face + dt_hair)
    final_f = feature_iresnet(P, "to_feature", torch.cat((f_face, f_hair), dim=1))
    return final_s, final_f


def post_process_param_shapes():
    """State-dict key -> shape of PostProcessModel (reference key order), + 'latent_avg'."""
    S = {}
    pre = "encoder_face"
    S[f"{pre}.conv.0.weight"] = (64, 3, 3, 3)
    _bn_shapes(S, f"{pre}.conv.1", 64)
    S[f"{pre}.conv.2.weight"] = (64,)
    inpl = 64
    for li, (planes, nblocks) in enumerate(IRESNET50_LAYERS):
        for j in range(nblocks):
            b = f"{pre}.block_{li + 1}.{j}"
            _bn_shapes(S, f"{b}.bn1", inpl)
            S[f"{b}.conv1.weight"] = (planes, inpl, 3, 3)
            _bn_shapes(S, f"{b}.bn2", planes)
            S[f"{b}.prelu.weight"] = (planes,)
            S[f"{b}.conv2.weight"] = (planes, planes, 3, 3)
            _bn_shapes(S, f"{b}.bn3", planes)
            if j == 0:
                S[f"{b}.downsample.0.weight"] = (planes, inpl, 1, 1)
                _bn_shapes(S, f"{b}.downsample.1", planes)
            inpl = planes
    cl = f"{pre}.content_layer.0"
    _bn_shapes(S, f"{cl}.0", 128)
    S[f"{cl}.1.weight"] = (512, 128, 3, 3)
    _bn_shapes(S, f"{cl}.2", 512)
    S[f"{cl}.3.weight"] = (512,)
    S[f"{cl}.4.weight"] = (512, 512, 3, 3)
    _bn_shapes(S, f"{cl}.5", 512)
    for i in range(18):
        S[f"{pre}.styles.{i}.weight"] = (512, 960 * 9)
        S[f"{pre}.styles.{i}.bias"] = (512,)
    inpl = 1024
    for n, (planes, num) in enumerate(FEATURE_BLOCKS, start=1):
        for k in range(1, num + 1):
            b = f"to_feature.res_blocks.res_block_{n}_{k}"
            _bn_shapes(S, f"{b}.bn1", inpl)
            S[f"{b}.conv1.weight"] = (planes, inpl, 3, 3)
            _bn_shapes(S, f"{b}.bn2", planes)
            S[f"{b}.prelu.weight"] = (planes,)
            S[f"{b}.conv2.weight"] = (planes, planes, 3, 3)
            _bn_shapes(S, f"{b}.bn3", planes)
            if inpl != planes:
                S[f"{b}.downsample.0.weight"] = (planes, inpl, 1, 1)
                _bn_shapes(S, f"{b}.downsample.1", planes)
            inpl = planes
    for name in ("to_latent_1", "to_latent_2"):
        for i in range(5):
            m = f"{name}.{i}"
            S[f"{m}.fc.weight"], S[f"{m}.fc.bias"] = (512, 512), (512,)
            for fn in ("gamma_function", "beta_function"):
                S[f"{m}.{fn}.0.weight"], S[f"{m}.{fn}.0.bias"] = (512, 512), (512,)
                S[f"{m}.{fn}.1.weight"], S[f"{m}.{fn}.1.bias"] = (512,), (512,)
                S[f"{m}.{fn}.3.weight"], S[f"{m}.{fn}.3.bias"] = (512, 512), (512,)
    S["latent_avg"] = (18, 512)
    return S


def _modulation_shapes(S, pre, inp=512, middle=512):
    S[f"{pre}.fc.weight"], S[f"{pre}.fc.bias"] = (512, 512), (512,)
    for name in ("gamma_function", "beta_function"):
        S[f"{pre}.{name}.0.weight"], S[f"{pre}.{name}.0.bias"] = (middle, inp), (middle,)
        S[f"{pre}.{name}.1.weight"], S[f"{pre}.{name}.1.bias"] = (middle,), (middle,)
        S[f"{pre}.{name}.3.weight"], S[f"{pre}.{name}.3.bias"] = (512, middle), (512,)


def rotate_param_shapes():
    """State-dict key -> shape of RotateModel (reference key order)."""
    S = {}
    for i in range(5):
        _modulation_shapes(S, f"modulation_module_list.{i}")
    return S


def clip_blending_param_shapes():
    """State-dict key -> shape of ClipBlendingModel's OWN parameters (the frozen `clip_model.*` entries excluded)."""
    S = {}
    for i in range(5):
        _modulation_shapes(S, f"modulation_module_list.{i}", inp=512 * 3, middle=1024)
    return S


def rotate_model(P, latent_from, latent_to):
    """RotateModel.forward (Encoders.py:66-71): [B,6,512] x 2 -> [B,6,512]."""
    dt = pixel_norm(latent_from)
    for i in range(5):
        dt = modulation_module(P, f"modulation_module_list.{i}", dt, latent_to, 6, i == 4)
    return latent_from + 0.1 * dt


CLIP_MEAN, CLIP_STD = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)


def clip_image_input(image):
    """What ClipBlendingModel.get_image_embed hands to clip_model.encode_image (Encoders.py:91-94):
    AdaptiveAvgPool2d(224) of the [-1,1] image, mapped to [0,1], CLIP-normalised."""
    x = F.adaptive_avg_pool2d(image, (224, 224)) * 0.5 + 0.5
    mean, std = torch.tensor(CLIP_MEAN).view(1, 3, 1, 1), torch.tensor(CLIP_STD).view(1, 3, 1, 1)
    return (x - mean) / std


def clip_blending(P, latent_face, latent_color, target_face, hair_color, image_embed):
    """ClipBlendingModel.forward (Encoders.py:96-103) with `image_embed` standing for clip_model.encode_image."""
    embed_face = image_embed(clip_image_input(target_face)).unsqueeze(1).expand(-1, 12, -1)
    embed_color = image_embed(clip_image_input(hair_color)).unsqueeze(1).expand(-1, 12, -1)
    latent_in = torch.cat((latent_color, embed_face, embed_color), dim=-1)
    dt = pixel_norm(latent_face)
    for i in range(5):
        dt = modulation_module(P, f"modulation_module_list.{i}", dt, latent_in, 12, i == 4)
    return latent_face + 0.1 * dt
