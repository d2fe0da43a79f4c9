"""CPU oracle: restatement of the two encoder forwards that feed the generator.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Functional restatements in plain
torch CPU fp32 ops, driven by flat state dicts with the reference's keys
(file:line relative to /root/reference):

  e4e   Encoder4Editing.forward     models/encoder4editing/models/encoders/psp_encoders.py:173-200
        bottleneck_IR_SE / SEModule  models/encoder4editing/models/encoders/helpers.py:57-120
        GradualStyleBlock            psp_encoders.py:34-55 (EqualLinear: models/stylegan2/model.py:128-157 of e4e's copy)
        _upsample_add                helpers.py:123-140
        get_latents (+ latent_avg)   models/encoder4editing/utils/model_utils.py:7-14
  FS    fs_encoder_v2.forward        models/FeatureStyleEncoder/nets/feature_style_encoder.py:47-65
        IBasicBlock / IResNet layers models/FeatureStyleEncoder/arcface/iresnet.py:28-57, 116-138
        downscale + dlatent_avg      models/FeatureStyleEncoder/trainer.py:61-64, 288-289

BatchNorm is evaluated in inference mode (running statistics), as the pipeline does
(`net.eval()`, model_utils.py:25; trainer.enc.eval()).  Pinned against the imported
reference by oracle/make_golden.py.
"""
import math

import torch
import torch.nn.functional as F

# get_blocks(50) of helpers.py:30-37: (in_channel, depth, stride) per unit
IR50_UNITS = ([(64, 64, 2)] + [(64, 64, 1)] * 2 + [(64, 128, 2)] + [(128, 128, 1)] * 3 +
              [(128, 256, 2)] + [(256, 256, 1)] * 13 + [(256, 512, 2)] + [(512, 512, 1)] * 2)
# iresnet50 of arcface/iresnet.py: blocks per layer, planes
IRESNET50_LAYERS = [(64, 3), (128, 4), (256, 14), (512, 3)]


def bn(P, pre, x, eps=1e-5):
    return F.batch_norm(x, P[f"{pre}.running_mean"], P[f"{pre}.running_var"], P[f"{pre}.weight"],
                        P[f"{pre}.bias"], False, 0.0, eps)


def ir_se_unit(P, pre, x, in_c, depth, stride):
    """bottleneck_IR_SE (helpers.py:93-120): BN -> conv3x3 -> PReLU -> conv3x3(stride) -> BN -> SE, + shortcut."""
    if in_c == depth:
        shortcut = x[:, :, ::stride, ::stride]  # MaxPool2d(1, stride)
    else:
        shortcut = bn(P, f"{pre}.shortcut_layer.1", F.conv2d(x, P[f"{pre}.shortcut_layer.0.weight"], stride=stride))
    r = bn(P, f"{pre}.res_layer.0", x)
    r = F.conv2d(r, P[f"{pre}.res_layer.1.weight"], padding=1)
    r = F.prelu(r, P[f"{pre}.res_layer.2.weight"])
    r = F.conv2d(r, P[f"{pre}.res_layer.3.weight"], stride=stride, padding=1)
    r = bn(P, f"{pre}.res_layer.4", r)
    g = r.mean(dim=(2, 3), keepdim=True)  # SEModule (helpers.py:57-73)
    g = F.relu(F.conv2d(g, P[f"{pre}.res_layer.5.fc1.weight"]))
    g = torch.sigmoid(F.conv2d(g, P[f"{pre}.res_layer.5.fc2.weight"]))
    return r * g + shortcut


def gradual_style_block(P, pre, x, n_convs=None):
    """GradualStyleBlock (psp_encoders.py:34-55): log2(spatial) x [conv3x3 s2 + LeakyReLU(0.01)] then EqualLinear."""
    i = 0
    while f"{pre}.convs.{2 * i}.weight" in P and (n_convs is None or i < n_convs):
        x = F.leaky_relu(F.conv2d(x, P[f"{pre}.convs.{2 * i}.weight"], P[f"{pre}.convs.{2 * i}.bias"], stride=2,
                                  padding=1), 0.01)
        i += 1
    w = P[f"{pre}.linear.weight"]
    x = x.reshape(-1, w.shape[1])
    return F.linear(x, w * (1.0 / math.sqrt(w.shape[1])), P[f"{pre}.linear.bias"])


def upsample_add(x, y):
    return F.interpolate(x, size=y.shape[2:], mode="bilinear", align_corners=True) + y


def e4e_forward(P, x, n_styles=18, latent_avg=None, return_taps=False):
    """Encoder4Editing(50, 'ir_se') at inference stage (all 17 deltas); x [B,3,256,256] in [-1,1]."""
    x = F.conv2d(x, P["input_layer.0.weight"], padding=1)
    x = F.prelu(bn(P, "input_layer.1", x), P["input_layer.2.weight"])
    taps = {}
    for i, (in_c, depth, stride) in enumerate(IR50_UNITS):
        x = ir_se_unit(P, f"body.{i}", x, in_c, depth, stride)
        if i in (6, 20, 23):
            taps[i] = x
    c1, c2, c3 = taps[6], taps[20], taps[23]
    w0 = gradual_style_block(P, "styles.0", c3)
    w = w0.unsqueeze(1).repeat(1, n_styles, 1)
    feats = c3
    for i in range(1, n_styles):
        if i == 3:
            feats = p2 = upsample_add(c3, F.conv2d(c2, P["latlayer1.weight"], P["latlayer1.bias"]))
        elif i == 7:
            feats = upsample_add(p2, F.conv2d(c1, P["latlayer2.weight"], P["latlayer2.bias"]))
        w[:, i] += gradual_style_block(P, f"styles.{i}", feats)
    if latent_avg is not None:  # get_latents, model_utils.py:9-13
        w = w + latent_avg.unsqueeze(0)
    return (w, (c1, c2, c3)) if return_taps else w


def ibasic_block(P, pre, x, stride):
    """IBasicBlock (iresnet.py:28-57)."""
    out = bn(P, f"{pre}.bn1", x)
    out = F.conv2d(out, P[f"{pre}.conv1.weight"], padding=1)
    out = F.prelu(bn(P, f"{pre}.bn2", out), P[f"{pre}.prelu.weight"])
    out = F.conv2d(out, P[f"{pre}.conv2.weight"], stride=stride, padding=1)
    out = bn(P, f"{pre}.bn3", out)
    if f"{pre}.downsample.0.weight" in P:
        x = bn(P, f"{pre}.downsample.1", F.conv2d(x, P[f"{pre}.downsample.0.weight"], stride=stride))
    return out + x


def fs_encoder_forward(P, x, n_styles=18, content_stride=2, dlatent_avg=None, return_features=False):
    """fs_encoder_v2.forward on a [B,3,256,256] input; returns (S [B,18,512], content [B,512,16,16])."""
    x = F.conv2d(x, P["conv.0.weight"], padding=1)
    x = F.prelu(bn(P, "conv.1", x), P["conv.2.weight"])
    pooled, content = [], None
    for li, (planes, nblocks) in enumerate(IRESNET50_LAYERS):
        for j in range(nblocks):
            x = ibasic_block(P, f"block_{li + 1}.{j}", x, 2 if j == 0 else 1)
        if li == 2:  # content branches off block_3's output (feature_style_encoder.py:57)
            c = bn(P, "content_layer.0", x)
            c = F.conv2d(c, P["content_layer.1.weight"], padding=1)
            c = F.prelu(bn(P, "content_layer.2", c), P["content_layer.3.weight"])
            c = F.conv2d(c, P["content_layer.4.weight"], stride=content_stride, padding=1)
            content = bn(P, "content_layer.5", c)
        pooled.append(F.adaptive_avg_pool2d(x, (3, 3)))
    flat = torch.cat(pooled, dim=1).reshape(x.shape[0], -1)
    s = torch.stack([F.linear(flat, P[f"styles.{i}.weight"], P[f"styles.{i}.bias"]) for i in range(n_styles)], dim=1)
    if dlatent_avg is not None:  # trainer.py:289
        s = s + dlatent_avg
    return (s, content, pooled) if return_features else (s, content)


def downscale2x_twice(x):
    """trainer.py:61-64 with scale=2, mode='bilinear': two F.interpolate(scale_factor=0.5)."""
    for _ in range(2):
        x = F.interpolate(x, scale_factor=0.5, mode="bilinear")
    return x


def fs_encoder_test(P, img_1024, dlatent_avg):
    """What Embedding.py:74-76 consumes from Trainer.test(img=..., return_latent=True):
    (w_recon [B,18,512], fea [B,512,16,16]).  The generator forward the reference also runs
    there (trainer.py:295) is discarded by the caller and is not part of this function."""
    return fs_encoder_forward(P, downscale2x_twice(img_1024), dlatent_avg=dlatent_avg)


def e4e_param_shapes():
    """State-dict key -> shape of Encoder4Editing(50, 'ir_se', stylegan_size=1024): 621 entries."""
    S = {"input_layer.0.weight": (64, 3, 3, 3)}
    _bn_shapes(S, "input_layer.1", 64)
    S["input_layer.2.weight"] = (64,)
    for i, (in_c, depth, stride) in enumerate(IR50_UNITS):
        pre = f"body.{i}"
        if in_c != depth:
            S[f"{pre}.shortcut_layer.0.weight"] = (depth, in_c, 1, 1)
            _bn_shapes(S, f"{pre}.shortcut_layer.1", depth)
        _bn_shapes(S, f"{pre}.res_layer.0", in_c)
        S[f"{pre}.res_layer.1.weight"] = (depth, in_c, 3, 3)
        S[f"{pre}.res_layer.2.weight"] = (depth,)
        S[f"{pre}.res_layer.3.weight"] = (depth, depth, 3, 3)
        _bn_shapes(S, f"{pre}.res_layer.4", depth)
        S[f"{pre}.res_layer.5.fc1.weight"] = (depth // 16, depth, 1, 1)
        S[f"{pre}.res_layer.5.fc2.weight"] = (depth, depth // 16, 1, 1)
    for i in range(18):
        spatial = 16 if i < 3 else (32 if i < 7 else 64)
        for k in range(int(math.log2(spatial))):
            S[f"styles.{i}.convs.{2 * k}.weight"] = (512, 512, 3, 3)
            S[f"styles.{i}.convs.{2 * k}.bias"] = (512,)
        S[f"styles.{i}.linear.weight"] = (512, 512)
        S[f"styles.{i}.linear.bias"] = (512,)
    S["latlayer1.weight"] = (512, 256, 1, 1)
    S["latlayer1.bias"] = (512,)
    S["latlayer2.weight"] = (512, 128, 1, 1)
    S["latlayer2.bias"] = (512,)
    return S


def fs_param_shapes(n_styles=18):
    """State-dict key -> shape of fs_encoder_v2 (iresnet50 trunk): 517 entries."""
    S = {"conv.0.weight": (64, 3, 3, 3)}
    _bn_shapes(S, "conv.1", 64)
    S["conv.2.weight"] = (64,)
    inpl = 64
    layers = {}
    for li, (planes, nblocks) in enumerate(IRESNET50_LAYERS):
        T = {}
        for j in range(nblocks):
            pre = f"block_{li + 1}.{j}"
            _bn_shapes(T, f"{pre}.bn1", inpl)
            T[f"{pre}.conv1.weight"] = (planes, inpl, 3, 3)
            _bn_shapes(T, f"{pre}.bn2", planes)
            T[f"{pre}.prelu.weight"] = (planes,)
            T[f"{pre}.conv2.weight"] = (planes, planes, 3, 3)
            _bn_shapes(T, f"{pre}.bn3", planes)
            if j == 0:
                T[f"{pre}.downsample.0.weight"] = (planes, inpl, 1, 1)
                _bn_shapes(T, f"{pre}.downsample.1", planes)
            inpl = planes
        layers[li] = T
    for li in range(4):
        S.update(layers[li])
    _bn_shapes(S, "content_layer.0", 256)
    S["content_layer.1.weight"] = (512, 256, 3, 3)
    _bn_shapes(S, "content_layer.2", 512)
    S["content_layer.3.weight"] = (512,)
    S["content_layer.4.weight"] = (512, 512, 3, 3)
    _bn_shapes(S, "content_layer.5", 512)
    for i in range(n_styles):
        S[f"styles.{i}.weight"] = (512, 960 * 9)
        S[f"styles.{i}.bias"] = (512,)
    return S


def _bn_shapes(S, pre, c):
    S[f"{pre}.weight"] = (c,)
    S[f"{pre}.bias"] = (c,)
    S[f"{pre}.running_mean"] = (c,)
    S[f"{pre}.running_var"] = (c,)
    S[f"{pre}.num_batches_tracked"] = ()
