"""TEST INFRASTRUCTURE - CPU restatement of the reference's SEAN inpainting stage.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (hairfastgan_amd/) never does.  Functional torch-CPU fp32 restatement of

  models/sean_codes/models/pix2pix_model.py:299-325   encode_sean / decode_sean (call site models/Alignment.py:126-131)
  .../pix2pix_model.py:121-148                         preprocess_input (label map -> one-hot)
  .../pix2pix_model.py:268-293                         load_average_feature (the per-label median style codes, data files
                                                       models/sean_codes/styles_test/mean_style_code/median/<label>/ACE.npy)
  models/sean_codes/models/networks/generator.py:72-110     SPADEGenerator.forward (num_upsampling_layers 'normal')
  .../networks/architecture.py:21-97                   SPADEResnetBlock (+ torch's spectral_norm in eval mode)
  .../networks/architecture.py:155-207                 Zencoder (style encoder + per-region average pooling)
  .../networks/normalization.py:70-208                 ACE (status 'UI_mode': per-region style codes -> fc_mu -> 3x3 convs,
                                                       blended with SPADE's gamma / beta), :211-257 SPADE

with SEAN_OPT (pix2pix_model.py:328-339: ngf 64, semantic_nc 19, crop_size 256, norm_G spectralspadesyncbatch3x3).
Pinned by oracle/make_golden.py against the imported reference (Pix2PixModel around its real SPADEGenerator, synthetic
parameters, `torch.randn(..., device='cuda')` of ACE.forward redirected to explicit noise, load_average_feature replaced
by synthetic codes).

Parameters: flat dict with the keys of `Pix2PixModel(SEAN_OPT).state_dict()` (`netG.Zencoder.model.1.weight`,
`netG.head_0.conv_0.weight_orig` / `.weight_u` / `.weight_v`, `netG.up_0.ace_s.fc_mu7.weight`, ...).

Noise: ACE adds `randn(B, W, H, 1) * noise_var` (transposed to [B, C, H, W], normalization.py:106) before its BatchNorm;
`noise` below is the list of those [B, W, H, 1] draws in call order (18 ACE calls per decode; None = zeros).
"""
import torch
import torch.nn.functional as F

N_LABELS = 19


class Cfg:
    """SEAN_OPT's sizes (ngf 64, crop_size 256; style_length 512 and nhidden 128 are hard-coded in normalization.py:80,
    :234; the Zencoder's widths in architecture.py:156-176).  Smaller values give the scaled-down model the CPU
    interpreter tests run (same code path, same state-dict layout)."""

    def __init__(self, ngf=64, style=512, hidden=128, size=256, zc=(32, 64, 128, 256)):
        self.ngf, self.style, self.hidden, self.size, self.zc = ngf, style, hidden, size, zc
        n = ngf
        # (name, fin, fout, use_rgb) in forward order (generator.py:35-43)
        self.blocks = [("head_0", 16 * n, 16 * n, True), ("G_middle_0", 16 * n, 16 * n, True), ("G_middle_1", 16 * n, 16 * n, True),
                       ("up_0", 16 * n, 8 * n, True), ("up_1", 8 * n, 4 * n, True), ("up_2", 4 * n, 2 * n, True),
                       ("up_3", 2 * n, 1 * n, False)]


DEFAULT = Cfg()


def ace_call_order(cfg=DEFAULT):
    """[(block, ace name, channels, spatial size)] in the order SPADEResnetBlock.forward calls them
    (architecture.py:67-93: the shortcut's ace_s first)."""
    out, res = [], cfg.size // 32
    for i, (name, fin, fout, _rgb) in enumerate(cfg.blocks):
        if i in (1, 3, 4, 5, 6):
            res *= 2
        fmid = min(fin, fout)
        if fin != fout:
            out.append((name, "ace_s", fin, res))
        out += [(name, "ace_0", fin, res), (name, "ace_1", fmid, res)]
    return out


def _ace_shapes(pre, norm_nc, use_rgb, cfg):
    N_HIDDEN, STYLE_LEN = cfg.hidden, cfg.style
    S = {f"{pre}.blending_gamma": (1,), f"{pre}.blending_beta": (1,), f"{pre}.noise_var": (norm_nc,)}
    for bn in (f"{pre}.Spade.param_free_norm", ):
        S[f"{bn}.running_mean"], S[f"{bn}.running_var"], S[f"{bn}.num_batches_tracked"] = (norm_nc,), (norm_nc,), ()
    S[f"{pre}.Spade.mlp_shared.0.weight"], S[f"{pre}.Spade.mlp_shared.0.bias"] = (N_HIDDEN, N_LABELS, 3, 3), (N_HIDDEN,)
    for nm in ("mlp_gamma", "mlp_beta"):
        S[f"{pre}.Spade.{nm}.weight"], S[f"{pre}.Spade.{nm}.bias"] = (norm_nc, N_HIDDEN, 3, 3), (norm_nc,)
    bn = f"{pre}.param_free_norm"
    S[f"{bn}.running_mean"], S[f"{bn}.running_var"], S[f"{bn}.num_batches_tracked"] = (norm_nc,), (norm_nc,), ()
    if use_rgb:
        for j in range(N_LABELS):
            S[f"{pre}.fc_mu{j}.weight"], S[f"{pre}.fc_mu{j}.bias"] = (STYLE_LEN, STYLE_LEN), (STYLE_LEN,)
        for nm in ("conv_gamma", "conv_beta"):
            S[f"{pre}.{nm}.weight"], S[f"{pre}.{nm}.bias"] = (norm_nc, STYLE_LEN, 3, 3), (norm_nc,)
    return S


def sean_param_shapes(prefix="netG.", cfg=DEFAULT):
    """State-dict layout (keys, shapes, order) of Pix2PixModel(SEAN_OPT) = its SPADEGenerator under `netG.`."""
    S = {}
    NGF = cfg.ngf
    z = "Zencoder.model"
    z0, z1, z2, z3 = cfg.zc
    for idx, (co, ci) in ((1, (z0, 3)), (4, (z1, z0)), (7, (z2, z1))):
        S[f"{z}.{idx}.weight"], S[f"{z}.{idx}.bias"] = (co, ci, 3, 3), (co,)
    S[f"{z}.10.weight"], S[f"{z}.10.bias"] = (z2, z3, 3, 3), (z3,)   # ConvTranspose2d: [cin, cout, k, k]
    S[f"{z}.14.weight"], S[f"{z}.14.bias"] = (cfg.style, z3, 3, 3), (cfg.style,)
    S["fc.weight"], S["fc.bias"] = (16 * NGF, N_LABELS, 3, 3), (16 * NGF,)
    for name, fin, fout, rgb in cfg.blocks:
        fmid = min(fin, fout)
        convs = [("conv_0", fmid, fin, 3, True), ("conv_1", fout, fmid, 3, True)] + ([("conv_s", fout, fin, 1, False)] if fin != fout else [])
        for cn, co, ci, k, bias in convs:   # torch.nn.utils.spectral_norm: bias, weight_orig, weight_u, weight_v
            if bias:
                S[f"{name}.{cn}.bias"] = (co,)
            S[f"{name}.{cn}.weight_orig"], S[f"{name}.{cn}.weight_u"], S[f"{name}.{cn}.weight_v"] = (co, ci, k, k), (co,), (ci * k * k,)
        S.update(_ace_shapes(f"{name}.ace_0", fin, rgb, cfg))
        S.update(_ace_shapes(f"{name}.ace_1", fmid, rgb, cfg))
        if fin != fout:
            S.update(_ace_shapes(f"{name}.ace_s", fin, rgb, cfg))
    S["conv_img.weight"], S["conv_img.bias"] = (3, NGF, 3, 3), (3,)
    return {prefix + k: v for k, v in S.items()}


def one_hot(labels, n=N_LABELS):  # pix2pix_model.py:133-140
    b, _, h, w = labels.shape
    return torch.zeros(b, n, h, w).scatter_(1, labels.long(), 1.0)


def sn_weight(P, pre):
    """torch.nn.utils.spectral_norm in eval mode: weight_orig / sigma, sigma = u . (W_mat v) with the stored u, v
    (no power iteration outside training)."""
    w = P[f"{pre}.weight_orig"]
    sigma = torch.dot(P[f"{pre}.weight_u"], torch.mv(w.reshape(w.shape[0], -1), P[f"{pre}.weight_v"]))
    return w / sigma


def _bn_eval(P, pre, x, eps=1e-5):  # SynchronizedBatchNorm2d(affine=False) in eval mode = F.batch_norm on running stats
    return F.batch_norm(x, P[f"{pre}.running_mean"], P[f"{pre}.running_var"], None, None, False, 0.1, eps)


def zencoder(P, image, seg, pre="netG.Zencoder.model"):  # architecture.py:155-207
    x = F.conv2d(F.pad(image, (1, 1, 1, 1), mode="reflect"), P[f"{pre}.1.weight"], P[f"{pre}.1.bias"])
    x = F.leaky_relu(F.instance_norm(x), 0.2)
    for idx in (4, 7):
        x = F.conv2d(x, P[f"{pre}.{idx}.weight"], P[f"{pre}.{idx}.bias"], stride=2, padding=1)
        x = F.leaky_relu(F.instance_norm(x), 0.2)
    x = F.conv_transpose2d(x, P[f"{pre}.10.weight"], P[f"{pre}.10.bias"], stride=2, padding=1, output_padding=1)
    x = F.leaky_relu(F.instance_norm(x), 0.2)
    codes = torch.tanh(F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), P[f"{pre}.14.weight"], P[f"{pre}.14.bias"]))
    seg = F.interpolate(seg, size=codes.shape[2:], mode="nearest")
    b, f = codes.shape[:2]
    out = torch.zeros(b, seg.shape[1], f)
    for i in range(b):
        for j in range(seg.shape[1]):
            m = seg[i, j].bool()
            area = int(m.sum())
            if area > 0:
                out[i, j] = codes[i].masked_select(m).reshape(f, area).mean(1)
    return out


def ace(P, pre, x, seg, codes, use_rgb, noise=None):
    """normalization.py:103-185, status 'UI_mode' generalised to a batch: sample i uses codes[i] ([19, 512]); regions
    absent from the (resized) segmentation map contribute nothing."""
    b, c, h, w = x.shape
    if noise is not None:
        x = x + (noise * P[f"{pre}.noise_var"]).transpose(1, 3)
    normalized = _bn_eval(P, f"{pre}.param_free_norm", x)
    seg = F.interpolate(seg, size=(h, w), mode="nearest")
    sp = f"{pre}.Spade"
    actv = F.relu(F.conv2d(seg, P[f"{sp}.mlp_shared.0.weight"], P[f"{sp}.mlp_shared.0.bias"], padding=1))
    gamma_spade = F.conv2d(actv, P[f"{sp}.mlp_gamma.weight"], P[f"{sp}.mlp_gamma.bias"], padding=1)
    beta_spade = F.conv2d(actv, P[f"{sp}.mlp_beta.weight"], P[f"{sp}.mlp_beta.bias"], padding=1)
    if not use_rgb:
        return normalized * (1 + gamma_spade) + beta_spade
    STYLE_LEN = codes.shape[-1]
    middle_avg = torch.zeros(b, STYLE_LEN, h, w)
    for i in range(b):
        for j in range(seg.shape[1]):
            m = seg[i, j].bool()
            area = int(m.sum())
            if area > 0:
                mu = F.relu(F.linear(codes[i, j], P[f"{pre}.fc_mu{j}.weight"], P[f"{pre}.fc_mu{j}.bias"]))
                middle_avg[i].masked_scatter_(m, mu.reshape(STYLE_LEN, 1).expand(STYLE_LEN, area))
    gamma_avg = F.conv2d(middle_avg, P[f"{pre}.conv_gamma.weight"], P[f"{pre}.conv_gamma.bias"], padding=1)
    beta_avg = F.conv2d(middle_avg, P[f"{pre}.conv_beta.weight"], P[f"{pre}.conv_beta.bias"], padding=1)
    ga, ba = torch.sigmoid(P[f"{pre}.blending_gamma"]), torch.sigmoid(P[f"{pre}.blending_beta"])
    gamma_final = ga * gamma_avg + (1 - ga) * gamma_spade
    beta_final = ba * beta_avg + (1 - ba) * beta_spade
    return normalized * (1 + gamma_final) + beta_final


def spade_resnet_block(P, pre, x, seg, codes, fin, fout, use_rgb, noise_iter):  # architecture.py:67-97
    if fin != fout:
        x_s = F.conv2d(ace(P, f"{pre}.ace_s", x, seg, codes, use_rgb, next(noise_iter)), sn_weight(P, f"{pre}.conv_s"))
    else:
        x_s = x
    dx = ace(P, f"{pre}.ace_0", x, seg, codes, use_rgb, next(noise_iter))
    dx = F.conv2d(F.leaky_relu(dx, 0.2), sn_weight(P, f"{pre}.conv_0"), P[f"{pre}.conv_0.bias"], padding=1)
    dx = ace(P, f"{pre}.ace_1", dx, seg, codes, use_rgb, next(noise_iter))
    dx = F.conv2d(F.leaky_relu(dx, 0.2), sn_weight(P, f"{pre}.conv_1"), P[f"{pre}.conv_1.bias"], padding=1)
    return x_s + dx


def spade_generator(P, seg, codes, noise=None, pre="netG", taps=None, cfg=DEFAULT):
    """generator.py:72-110 with per-sample style codes [B, 19, 512] (what `obj_dic[str(j)]['ACE']` holds in UI_mode)."""
    n_ace = len(ace_call_order(cfg))
    it = iter(noise if noise is not None else [None] * n_ace)
    sw = cfg.size // 32  # compute_latent_vector_size (:55-70), 5 up-sampling layers
    x = F.conv2d(F.interpolate(seg, size=(sw, sw)), P[f"{pre}.fc.weight"], P[f"{pre}.fc.bias"], padding=1)
    for i, (name, fin, fout, rgb) in enumerate(cfg.blocks):
        if i in (1, 3, 4, 5, 6):
            x = F.interpolate(x, scale_factor=2)  # nn.Upsample(scale_factor=2): nearest
        x = spade_resnet_block(P, f"{pre}.{name}", x, seg, codes, fin, fout, rgb, it)
        if taps is not None:
            taps[name] = x
    x = F.conv2d(F.leaky_relu(x, 0.2), P[f"{pre}.conv_img.weight"], P[f"{pre}.conv_img.bias"], padding=1)
    return torch.tanh(x)


def encode_sean(P, images, labels):  # pix2pix_model.py:299-307: mode 'style_code'
    return zencoder(P, images, one_hot(labels))


def merge_codes(image_code, mean_codes):
    """decode_sean's obj_dic (pix2pix_model.py:311-316): the image's code where it is not all zero (= the label occurs
    in the image), the median code of the label otherwise.  image_code [B,19,512], mean_codes [19,512]."""
    absent = (image_code == 0).all(dim=-1, keepdim=True)
    return torch.where(absent, mean_codes.unsqueeze(0).expand_as(image_code), image_code)


def decode_sean(P, image_code, target_mask, mean_codes, noise=None, cfg=DEFAULT):
    """pix2pix_model.py:310-325 for B codes [B,19,512] and B (or 1, shared) target label maps [B|1,1,256,256]:
    returns [B,3,256,256] in (-1, 1)."""
    codes = merge_codes(image_code, mean_codes)
    seg = one_hot(target_mask)
    if seg.shape[0] == 1 and codes.shape[0] > 1:
        seg = seg.expand(codes.shape[0], -1, -1, -1)
    return spade_generator(P, seg, codes, noise, cfg=cfg)


def sean_inpaint(P, images_256, labels, target_mask, mean_codes, noise1=None, noise2=None):
    """Alignment.py:126-131: the two images of a pair re-rendered on the target mask."""
    codes = encode_sean(P, images_256, labels)
    g1 = decode_sean(P, codes[0:1], target_mask, mean_codes, noise1)[0]
    g2 = decode_sean(P, codes[1:2], target_mask, mean_codes, noise2)[0]
    return [g1, g2]
