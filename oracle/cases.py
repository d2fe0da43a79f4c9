"""Shared case tables + input builders for golden vectors (TEST INFRASTRUCTURE).

Used by oracle/make_golden.py (which runs the real reference on them) and by
tests/ (which run the oracle and the HIP path on the very same inputs).  All
inputs are formulas (oracle.synth), nothing is stored but outputs.
"""
import numpy as np
import torch

from . import synth

UPFIRDN_CASES = {
    "blur_2x5x9x9": dict(shape=(2, 5, 9, 9), up=1, down=1, pad=(1, 1)),
    "blur_1x2x65x65": dict(shape=(1, 2, 65, 65), up=1, down=1, pad=(1, 1)),
    "blur_1x3x33x17": dict(shape=(1, 3, 33, 17), up=1, down=1, pad=(1, 1)),
    "blur_pad21_1x4x8x8": dict(shape=(1, 4, 8, 8), up=1, down=1, pad=(2, 1)),
    "up2_1x3x7x7": dict(shape=(1, 3, 7, 7), up=2, down=1, pad=(2, 1)),
    "up2_2x3x16x16": dict(shape=(2, 3, 16, 16), up=2, down=1, pad=(2, 1)),
    "up2_1x3x64x64": dict(shape=(1, 3, 64, 64), up=2, down=1, pad=(2, 1)),
    "down2_1x2x10x10": dict(shape=(1, 2, 10, 10), up=1, down=2, pad=(1, 1)),
}

ACT_CASES = {"act_2x7x5x5": (2, 7, 5, 5), "act_3x512": (3, 512), "act_1x32x16x16": (1, 32, 16, 16)}

# name, cin, cout, style_dim, B, H, W
MODCONV_SMALL = [
    ("s16to8_h8", 16, 8, 32, 2, 8, 8),
    ("s32to32_h16", 32, 32, 64, 3, 16, 16),
    ("s64to32_h12x20", 64, 32, 512, 1, 12, 20),
]

# tag -> (size, channel_multiplier, n_mlp, batches, ranges)
GENERATOR_CASES = {
    "g64": (64, 2, 2, [2], [(0, 4), (0, 2), (2, 2), (3, 4)]),
    "g1024": (1024, 2, 8, [1, 2], [(0, 8), (0, 3), (3, 3), (4, 8), (5, 8)]),
}


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def blur_kernel4():
    return t(synth.fill_value("x.blur.kernel", (4, 4)))


def upfirdn_input(name):
    return t(synth.pseudo_normal(f"upfirdn/{name}", UPFIRDN_CASES[name]["shape"]))


def act_inputs(name):
    shape = ACT_CASES[name]
    return (t(synth.pseudo_normal(f"act/{name}", shape)),
            t(synth.unit_uniform(f"act/{name}/bias", (shape[1],))))


def styled_param_shapes(cin, cout, sdim, up):
    S = {"conv.weight": (1, cout, cin, 3, 3)}
    if up:
        S["conv.blur.kernel"] = (4, 4)
    S.update({"conv.modulation.weight": (cin, sdim), "conv.modulation.bias": (cin,),
              "noise.weight": (1,), "activate.bias": (cout,)})
    return S


def rgb_param_shapes(cin, sdim):
    return {"bias": (1, 3, 1, 1), "upsample.kernel": (4, 4), "conv.weight": (1, 3, cin, 1, 1),
            "conv.modulation.weight": (cin, sdim), "conv.modulation.bias": (cin,)}


def small_params(pre, shapes):
    """Parameters for one small module; keys prefixed 'L.' (oracle prefix)."""
    P = {}
    for k, shp in shapes.items():
        key = k if k.endswith("kernel") else f"{pre}.{k}"
        P[f"L.{k}"] = t(synth.fill_value(key, shp))
    return P


def modconv_small_inputs(name):
    for n, cin, cout, sdim, B, H, W in MODCONV_SMALL:
        if n == name:
            break
    else:
        raise KeyError(name)
    d = dict(cin=cin, cout=cout, sdim=sdim, B=B, H=H, W=W)
    d["x"] = t(synth.pseudo_normal(f"mc/{name}/x", (B, cin, H, W)))
    d["w"] = t(synth.pseudo_normal(f"mc/{name}/w", (B, sdim)))
    for up in (False, True):
        pre = f"mc/{name}/up{int(up)}"
        d[f"P_up{int(up)}"] = small_params(pre, styled_param_shapes(cin, cout, sdim, up))
        oh, ow = (2 * H, 2 * W) if up else (H, W)
        d[f"noise_up{int(up)}"] = t(synth.pseudo_normal(f"{pre}/noise", (B, 1, oh, ow)))
    pre = f"mc/{name}/rgb"
    d["P_rgb"] = small_params(pre, rgb_param_shapes(cout, sdim))
    d["x_rgb"] = t(synth.pseudo_normal(f"{pre}/x", (B, cout, 2 * H, 2 * W)))
    d["skip"] = t(synth.pseudo_normal(f"{pre}/skip", (B, 3, H, W)))
    return d


def generator_params(shapes):
    return {k: t(v) for k, v in synth.fill_state_dict(shapes).items()}


def generator_inputs(size, B, start_layer, cin_of_block=None):
    """latent W+, explicit noise list, and layer_in for `start_layer` (or None)."""
    log_size = int(np.log2(size))
    lat = t(synth.latent_wplus(B, n_latent=2 * log_size - 2))
    noise = [t(a) for a in synth.noise_maps(log_size=log_size)]
    layer_in = None
    if start_layer > 0:
        r = 2 ** (start_layer + 1)  # block s consumes 2^(s+1), produces 2^(s+2)
        layer_in = t(synth.pseudo_normal(f"layer_in/{start_layer}/{B}", (B, cin_of_block, r, r)))
    return lat, noise, layer_in


def edge_crops(size, c=32):
    """Crops of a [.., size, size] image away from its centre: the four corners and the middle of
    each edge (tile-edge coverage the centre crop does not give).  name -> (slice_y, slice_x)."""
    m = size // 2 - c // 2
    lo, hi, mid = slice(0, c), slice(size - c, size), slice(m, m + c)
    return {"tl": (lo, lo), "tr": (lo, hi), "bl": (hi, lo), "br": (hi, hi),
            "top": (lo, mid), "bottom": (hi, mid), "left": (mid, lo), "right": (mid, hi)}


def mapping_inputs(B=3, dim=512):
    """z for the mapping network (Generator.style / input_is_latent=False)."""
    return t(synth.pseudo_normal(f"mapping/z/{B}", (B, dim)))


# ------------------------------------------------------------------------------------
# encoders (oracle/ref_encoders.py)
# ------------------------------------------------------------------------------------
# name -> (in_c, depth, stride, B, H, W)
IRSE_UNIT_CASES = {"irse_16to32_s2": (16, 32, 2, 2, 12, 12), "irse_32to32_s1": (32, 32, 1, 1, 10, 14),
                   "irse_32to32_s2": (32, 32, 2, 2, 9, 16)}
IBASIC_CASES = {"ibasic_16to32_s2": (16, 32, 2, 2, 12, 12), "ibasic_32to32_s1": (32, 32, 1, 1, 10, 14)}
# name -> (channels, spatial, B)
STYLE_BLOCK_CASES = {"gsb_32_sp4": (32, 4, 3), "gsb_64_sp8": (64, 8, 2)}


def params_from_shapes(prefix, shapes):
    return {k: t(synth.fill_value(f"{prefix}.{k}", tuple(s))) for k, s in shapes.items()}


def irse_unit_shapes(in_c, depth):
    S = {}
    if in_c != depth:
        S["u.shortcut_layer.0.weight"] = (depth, in_c, 1, 1)
        for k in ("weight", "bias", "running_mean", "running_var"):
            S[f"u.shortcut_layer.1.{k}"] = (depth,)
    for k in ("weight", "bias", "running_mean", "running_var"):
        S[f"u.res_layer.0.{k}"] = (in_c,)
        S[f"u.res_layer.4.{k}"] = (depth,)
    S["u.res_layer.1.weight"] = (depth, in_c, 3, 3)
    S["u.res_layer.2.weight"] = (depth,)
    S["u.res_layer.3.weight"] = (depth, depth, 3, 3)
    S["u.res_layer.5.fc1.weight"] = (depth // 16, depth, 1, 1)
    S["u.res_layer.5.fc2.weight"] = (depth, depth // 16, 1, 1)
    return S


def ibasic_shapes(in_c, planes, stride):
    S = {}
    for name, c in (("bn1", in_c), ("bn2", planes), ("bn3", planes)):
        for k in ("weight", "bias", "running_mean", "running_var"):
            S[f"u.{name}.{k}"] = (c,)
    S["u.conv1.weight"] = (planes, in_c, 3, 3)
    S["u.prelu.weight"] = (planes,)
    S["u.conv2.weight"] = (planes, planes, 3, 3)
    if stride != 1 or in_c != planes:
        S["u.downsample.0.weight"] = (planes, in_c, 1, 1)
        for k in ("weight", "bias", "running_mean", "running_var"):
            S[f"u.downsample.1.{k}"] = (planes,)
    return S


def style_block_shapes(c, spatial):
    S = {}
    for k in range(int(np.log2(spatial))):
        S[f"u.convs.{2 * k}.weight"] = (c, c, 3, 3)
        S[f"u.convs.{2 * k}.bias"] = (c,)
    S["u.linear.weight"] = (c, c)
    S["u.linear.bias"] = (c,)
    return S


def unit_input(name, shape):
    return t(synth.pseudo_normal(f"enc/{name}/x", shape))


def e4e_inputs(B=2):
    x = t(synth.pseudo_normal(f"enc/e4e/x/{B}", (B, 3, 256, 256))) * 0.5
    latent_avg = t(synth.pseudo_normal("enc/e4e/latent_avg", (18, 512))) * 0.1
    return x, latent_avg


def fs_inputs(B=2):
    img = t(synth.pseudo_normal(f"enc/fs/img/{B}", (B, 3, 1024, 1024))) * 0.5
    dlatent_avg = t(synth.pseudo_normal("enc/fs/dlatent_avg", (18, 512))) * 0.1
    return img, dlatent_avg


def pp_inputs():
    """PostProcessModel inputs (Blending.py:66): source = the face image, target = the blended image,
    both normalised [1,3,256,256]."""
    src = t(synth.pseudo_normal("pp/source", (1, 3, 256, 256))) * 0.5
    tgt = t(synth.pseudo_normal("pp/target", (1, 3, 256, 256))) * 0.5
    return src, tgt


def latent_model_inputs():
    """RotateModel / ClipBlendingModel inputs (Alignment.py:61, Blending.py:60): W+ rows [2,6,512] x 2;
    S rows [2,12,512] x 2 and two masked [2,3,256,256] images in [-1,1]."""
    w_from = t(synth.pseudo_normal("latent/rot/from", (2, 6, 512)))
    w_to = t(synth.pseudo_normal("latent/rot/to", (2, 6, 512)))
    s_face = t(synth.pseudo_normal("latent/blend/face", (2, 12, 512)))
    s_color = t(synth.pseudo_normal("latent/blend/color", (2, 12, 512)))
    img_face = (t(synth.pseudo_normal("latent/blend/img_face", (2, 3, 256, 256))) * 0.4).clamp(-1, 1)
    img_color = (t(synth.pseudo_normal("latent/blend/img_color", (2, 3, 256, 256))) * 0.4).clamp(-1, 1)
    return w_from, w_to, s_face, s_color, img_face, img_color


def fake_clip_embed(x):
    """Deterministic stand-in for clip_model.encode_image in tests and golden generation ([B,3,224,224] -> [B,512]):
    the CLIP tower is an un-vendored dependency of the reference; what is pinned is everything around it."""
    return x.flatten(1)[:, ::294][:, :512].contiguous() * 0.5


def shape_masks():
    """Two pairs of CelebAMask-style label maps, long [2,1,256,256] each (Alignment.py:74-75: the face image's mask and the
    rotated shape image's mask): background, a skin ellipse with a few facial regions, neck / cloth, and a hair region
    (label 13) whose outline differs between the two maps and the two pairs."""
    yy, xx = torch.meshgrid(torch.arange(256, dtype=torch.float32), torch.arange(256, dtype=torch.float32), indexing="ij")

    def one(cx, cy, hair_w, hair_top, tilt):
        m = torch.zeros(256, 256, dtype=torch.long)
        m[(yy > 200) & ((xx - cx).abs() < 90)] = 18                                              # cloth
        m[(yy > 170) & (yy <= 215) & ((xx - cx).abs() < 35)] = 17                                # neck
        hair = (((xx - cx - tilt * (yy - cy) * 0.2) / hair_w) ** 2 + ((yy - cy + 25) / (cy - hair_top)) ** 2 < 1) & (yy < cy + 60)
        m[hair] = 13
        m[((xx - cx) / 52) ** 2 + ((yy - cy) / 70) ** 2 < 1] = 1                                 # skin
        for lab, ex, ey, rw, rh in ((4, -20, -15, 9, 5), (5, 20, -15, 9, 5), (6, -20, -28, 12, 3), (7, 20, -28, 12, 3),
                                    (2, 0, 8, 8, 14), (11, 0, 35, 16, 5), (12, 0, 30, 14, 3), (13 - 13 + 10, 0, 38, 12, 3)):
            m[((xx - cx - ex) / rw) ** 2 + ((yy - cy - ey) / rh) ** 2 < 1] = lab
        return m

    a = torch.stack([one(128, 120, 75, 25, 0.0), one(120, 125, 85, 35, 0.6)])[:, None]
    b = torch.stack([one(132, 118, 95, 15, -0.5), one(128, 122, 70, 40, 0.2)])[:, None]
    return a, b


def shape_adaptor_params():
    """Synthetic parameters of the CtrlHair mask generator: the closed-form fill with Linear weights scaled by
    1/sqrt(fan_in), LayerNorm gains in [0.25, 1], small shifts, and pseudo-random class-score convs (so that the argmax
    map has several regions and a spread of top-1 / top-2 margins)."""
    from . import ref_shape_adaptor as SA

    P = params_from_shapes("shape_adaptor", SA.param_shapes())
    for k in list(P):
        if k.endswith(".fc.weight"):
            P[k] = P[k] / float(P[k].shape[1]) ** 0.5
        elif k.endswith(".gamma"):
            P[k] = 0.25 + 0.75 * P[k].abs()
        elif k.endswith(".beta"):
            P[k] = P[k] * 0.2
    for name in ("hair_decoder", "face_decoder"):
        k = f"{name}.out_layer.conv.weight"
        P[k] = t(synth.pseudo_normal(f"shape_adaptor/{k}", tuple(P[k].shape))) * 0.3
    return P


def bisenet_input(tag):
    """ImageNet-normalised image for BiSeNet / get_segmentation: "512" = the [1,3,512,512] case of
    Embedding.py:81 (a smooth pattern + noise so that the argmax map has regions, not salt and pepper),
    "320x384" = a non-square plane (odd feature-map sizes on the way down)."""
    h, w = (512, 512) if tag == "512" else (320, 384)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, h), torch.linspace(-1, 1, w), indexing="ij")
    base = torch.stack([torch.sin(3 * xx + yy), torch.cos(2 * yy - xx), xx * yy])[None]
    return (base * 1.5 + 0.3 * t(synth.pseudo_normal(f"bisenet/x/{tag}", (1, 3, h, w)))).contiguous()


def bisenet_params():
    """Synthetic BiSeNet parameters: the closed-form fill, except for the three class-score convs, whose rows would
    otherwise be near-copies of each other (one class wins everywhere): pseudo-random rows give an argmax map with
    many regions and a spread of top-1 / top-2 margins, which is what the mask-index parity check needs."""
    from . import ref_bisenet as BS

    P = params_from_shapes("bisenet", BS.bisenet_param_shapes())
    for k in list(P):
        if P[k].ndim == 4:                                    # conv gains x3: activations of O(1) that follow the
            P[k] = P[k] * 3.0                                 # image instead of the BatchNorm shifts
        elif k.endswith(".bias") or k.endswith("running_mean"):
            P[k] = P[k] * 0.1
    for name in ("conv_out", "conv_out16", "conv_out32"):
        k = f"{name}.conv_out.weight"
        P[k] = t(synth.pseudo_normal(f"bisenet/{k}", tuple(P[k].shape))) * 0.5
    return P


# ------------------------------------------------------------------------------------
# SEAN inpainting (oracle/ref_sean.py)
# ------------------------------------------------------------------------------------
def sean_params(cfg=None):
    """Synthetic Pix2PixModel(SEAN_OPT) parameters: the closed-form fill with (i) spectral-norm triples whose u is one
    power-iteration step from a random unit v, times a gain that keeps the activations O(1) through the seven blocks
    (sigma = gain * |W v| > 0, computed in float64 so that every machine rounds to the same fp32 values; a random W has
    |W v| ~ its spectral norm / (1 + sqrt(n/m)), which would amplify 3x per conv), (ii) fc_mu weights scaled by
    1/sqrt(512), (iii) small ACE noise strengths, (iv) a conv_img gain that leaves the final tanh unsaturated."""
    from . import ref_sean as SN

    shapes = SN.sean_param_shapes(cfg=cfg or SN.DEFAULT)
    P = {}
    for k, shp in shapes.items():
        leaf = k.rsplit(".", 1)[-1]
        if leaf == "weight_orig":
            fan_in = shp[1] * shp[2] * shp[3]
            P[k] = t(synth.unit_uniform(k, shp) * np.float32(0.7 / np.sqrt(fan_in)))
        elif leaf == "weight_v":
            v = synth.pseudo_normal(k, shp).astype(np.float64)
            P[k] = t((v / np.linalg.norm(v)).astype(np.float32))
        elif leaf == "weight_u":
            continue  # below, from W and v
        elif ".fc_mu" in k and leaf == "weight":
            P[k] = t(synth.unit_uniform(k, shp) * np.float32(1.0 / np.sqrt(shp[1])))
        elif leaf == "noise_var":
            P[k] = t(np.float32(0.1) * synth.unit_uniform(k, shp))
        elif leaf == "num_batches_tracked":
            P[k] = torch.zeros((), dtype=torch.int64)
        else:
            P[k] = t(synth.fill_value(k, shp))
    for k in shapes:
        if k.endswith("weight_u"):
            pre = k[:-len("weight_u")]
            w = P[pre + "weight_orig"].numpy().astype(np.float64)
            wv = w.reshape(w.shape[0], -1) @ P[pre + "weight_v"].numpy().astype(np.float64)
            gain = 6.0 if ".conv_1." in k else (3.0 if ".conv_0." in k else 1.5)
            P[k] = t((gain * wv / np.linalg.norm(wv)).astype(np.float32))
    P["netG.conv_img.weight"] = P["netG.conv_img.weight"] * 0.5
    return {k: P[k] for k in shapes}


def sean_mean_codes():
    """Stand-in for the per-label median style codes (models/sean_codes/styles_test/mean_style_code/median/*/ACE.npy)."""
    return t(synth.pseudo_normal("sean/mean_codes", (19, 512))) * 0.5


def sean_inputs():
    """One pair as Alignment.align_images hands it to SEAN (Alignment.py:123-131): images [2,3,256,256] in [0,1], their label
    maps long [2,1,256,256] (the second one lacks a few labels, so decode_sean's median-code rule is exercised), the
    target label map [1,1,256,256], and the explicit ACE noise of the two decodes (18 draws [1,W,H,1] each)."""
    from . import ref_sean as SN

    images = t(synth.uniform01("sean/images", (2, 3, 256, 256)))
    m1, m2 = shape_masks()
    labels = torch.stack([m1[0], m2[0]])
    labels[1][labels[1] == 2] = 1
    labels[1][labels[1] == 18] = 0
    target = m1[1:2].clone()
    noise = [[t(synth.pseudo_normal(f"sean/noise/{d}/{i}", (1, r, r, 1))) for i, (_b, _a, _c, r) in enumerate(SN.ace_call_order())]
             for d in range(2)]
    return images, labels, target, noise


# ------------------------------------------------------------------------------------
# CLIP ViT image tower (oracle/ref_clip.py)
# ------------------------------------------------------------------------------------
def clip_params(tag="clip", **sizes):
    """Synthetic `visual.*` parameters of the CLIP image tower: matrices scaled by 1/sqrt(fan_in), LayerNorm gains in
    [0.5, 1.5], small biases / embeddings."""
    from . import ref_clip as RC

    P = {}
    for k, shp in RC.clip_visual_param_shapes(**sizes).items():
        leaf = k.rsplit(".", 1)[-1]
        if len(shp) == 2 and leaf != "positional_embedding":
            P[k] = t(synth.unit_uniform(f"{tag}.{k}", shp)) * (1.0 / shp[-1 if leaf != "proj" else 0] ** 0.5)
        elif len(shp) == 4:
            P[k] = t(synth.unit_uniform(f"{tag}.{k}", shp)) * (1.0 / (shp[1] * shp[2] * shp[3]) ** 0.5)
        elif "ln_" in k and leaf == "weight":
            P[k] = 0.5 + t(synth.uniform01(f"{tag}.{k}", shp))
        else:
            P[k] = t(synth.unit_uniform(f"{tag}.{k}", shp)) * 0.2
    return P


# ------------------------------------------------------------------------------------
# one complete swap (oracle/make_pipeline_golden.py, tests/test_gpu_pipeline.py)
# ------------------------------------------------------------------------------------
def pipeline_images():
    """face / shape / color: three uint8 [3,1024,1024] images - smooth patterns plus noise (so that the synthetic BiSeNet
    yields maps with regions; after ImageNet normalisation the pattern has the amplitude of bisenet_input)."""
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, 1024), torch.linspace(-1, 1, 1024), indexing="ij")
    out = []
    for k, (a, b, c) in enumerate(((3.0, 1.0, 2.0), (2.5, -1.5, 3.0), (3.5, 0.5, -2.5))):
        base = torch.stack([torch.sin(a * xx + b * yy), torch.cos(c * yy - b * xx), torch.sin(a * xx * yy + c)])
        img = 0.5 + 0.33 * base + 0.07 * t(synth.pseudo_normal(f"pipeline/img/{k}", (3, 1024, 1024)))
        out.append((img.clamp(0, 1) * 255).round().to(torch.uint8))
    return tuple(out)


def pipeline_sean_noise(k, r):
    """The k-th ACE noise draw of the swap's SEAN decodes (decode = k // 18, ACE = k % 18), natural [1, H, W] order."""
    return t(synth.pseudo_normal(f"pipeline/sean_noise/{k}", (1, r, r)))


def pipeline_bisenet_params():
    """bisenet_params() with the class-score rows of 'nose' (BiSeNet class 10) and 'hair' (17) exchanged: the synthetic
    network then labels about a third of every image as hair (CelebA index 13), so that the swap's hair-mask algebra,
    mixing and F-space alignment are exercised (with the plain synthetic parameters no pixel is hair)."""
    P = dict(bisenet_params())
    w = P["conv_out.conv_out.weight"].clone()
    w[[10, 17]] = w[[17, 10]]
    P["conv_out.conv_out.weight"] = w
    return P
