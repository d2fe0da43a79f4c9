"""TEST INFRASTRUCTURE - CPU restatement of the reference's face parsing.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (hairfastgan_amd/) never does.  Functional torch-CPU fp32 restatement of

  models/CtrlHair/external_code/face_parsing/model.py:230-253   BiSeNet.forward (first output only is used)
  .../model.py:92-127 ContextPath, :68-89 AttentionRefinementModule, :178-227 FeatureFusionModule,
  .../model.py:13-53 ConvBNReLU / BiSeNetOutput
  .../resnet.py:19-82 BasicBlock / Resnet18 (torchvision layout: conv1 7x7/2, maxpool 3x3/2, 4 x 2 blocks)
  .../my_parsing_util.py:72-95 FaceParsing_tensor.parsing_img (argmax over the 19 classes) and
      swap_parsing_label_to_celeba_mask (label permutation PARSING_LABEL_LIST, global_value_utils.py:49-51)
  models/Net.py:108-115 get_segmentation (+ nearest resize to 256x256)

pinned by oracle/make_golden.py against the imported reference (its Resnet18 constructor downloads weights:
`torch.utils.model_zoo.load_url` is replaced by a function returning synthetic tensors of the right shapes).
Parameters: flat dict with the reference's state-dict keys (`cp.resnet.*`, `cp.arm16.*`, `ffm.*`, `conv_out.*`...).
"""
import torch
import torch.nn.functional as F

from .ref_encoders import _bn_shapes, bn

# FaceParsing_tensor.label_list (my_parsing_util.py:58-62) and PARSING_LABEL_LIST (global_value_utils.py:49-51)
BISENET_LABELS = ["background", "skin_other", "l_brow", "r_brow", "l_eye", "r_eye", "eye_g", "l_ear", "r_ear", "ear_r",
                  "nose", "mouth", "u_lip", "l_lip", "neck", "neck_l", "cloth", "hair", "hat"]
CELEBA_LABELS = ["background", "skin_other", "nose", "eye_g", "l_eye", "r_eye", "l_brow", "r_brow", "l_ear", "r_ear",
                 "mouth", "u_lip", "l_lip", "hair", "hat", "ear_r", "neck_l", "neck", "cloth"]
# remap[bisenet class] = CelebAMask-style index (13 = hair): what swap_parsing_label_to_celeba_mask computes
LABEL_REMAP = [CELEBA_LABELS.index(n) for n in BISENET_LABELS]


def conv_bn_relu(P, pre, x, stride=1, padding=1):  # model.py:13-29
    return F.relu(bn(P, f"{pre}.bn", F.conv2d(x, P[f"{pre}.conv.weight"], stride=stride, padding=padding)))


def basic_block(P, pre, x, stride):  # resnet.py:19-46
    r = F.relu(bn(P, f"{pre}.bn1", F.conv2d(x, P[f"{pre}.conv1.weight"], stride=stride, padding=1)))
    r = bn(P, f"{pre}.bn2", F.conv2d(r, P[f"{pre}.conv2.weight"], padding=1))
    sc = x
    if f"{pre}.downsample.0.weight" in P:
        sc = bn(P, f"{pre}.downsample.1", F.conv2d(x, P[f"{pre}.downsample.0.weight"], stride=stride))
    return F.relu(sc + r)


def resnet18(P, pre, x):  # resnet.py:56-77
    x = F.relu(bn(P, f"{pre}.bn1", F.conv2d(x, P[f"{pre}.conv1.weight"], stride=2, padding=3)))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    feats = []
    for li, stride in ((1, 1), (2, 2), (3, 2), (4, 2)):
        x = basic_block(P, f"{pre}.layer{li}.0", x, stride)
        x = basic_block(P, f"{pre}.layer{li}.1", x, 1)
        feats.append(x)
    return feats[1], feats[2], feats[3]  # 1/8, 1/16, 1/32


def arm(P, pre, x):  # model.py:68-89
    feat = conv_bn_relu(P, f"{pre}.conv", x)
    atten = F.avg_pool2d(feat, feat.shape[2:])
    atten = torch.sigmoid(bn(P, f"{pre}.bn_atten", F.conv2d(atten, P[f"{pre}.conv_atten.weight"])))
    return feat * atten


def context_path(P, pre, x):  # model.py:92-127
    feat8, feat16, feat32 = resnet18(P, f"{pre}.resnet", x)
    avg = F.avg_pool2d(feat32, feat32.shape[2:])
    avg = conv_bn_relu(P, f"{pre}.conv_avg", avg, padding=0)
    avg_up = F.interpolate(avg, feat32.shape[2:], mode="nearest")
    feat32_up = F.interpolate(arm(P, f"{pre}.arm32", feat32) + avg_up, feat16.shape[2:], mode="nearest")
    feat32_up = conv_bn_relu(P, f"{pre}.conv_head32", feat32_up)
    feat16_up = F.interpolate(arm(P, f"{pre}.arm16", feat16) + feat32_up, feat8.shape[2:], mode="nearest")
    feat16_up = conv_bn_relu(P, f"{pre}.conv_head16", feat16_up)
    return feat8, feat16_up, feat32_up


def ffm(P, pre, fsp, fcp):  # model.py:178-207
    feat = conv_bn_relu(P, f"{pre}.convblk", torch.cat([fsp, fcp], dim=1), padding=0)
    atten = F.avg_pool2d(feat, feat.shape[2:])
    atten = torch.sigmoid(F.conv2d(F.relu(F.conv2d(atten, P[f"{pre}.conv1.weight"])), P[f"{pre}.conv2.weight"]))
    return feat * atten + feat


def bisenet_logits(P, x):
    """BiSeNet.forward(x)[0]: [B,3,H,W] ImageNet-normalised -> [B,19,H,W] (model.py:239-253; the two
    auxiliary outputs are unused by HairFast)."""
    H, W = x.shape[2:]
    feat_res8, feat_cp8, _ = context_path(P, "cp", x)
    fuse = ffm(P, "ffm", feat_res8, feat_cp8)
    out = F.conv2d(conv_bn_relu(P, "conv_out.conv", fuse), P["conv_out.conv_out.weight"])
    return F.interpolate(out, (H, W), mode="bilinear", align_corners=True)


def get_segmentation(P, img_rgb, resize=True):
    """models/Net.py:108-115: argmax of the logits of ONE image, remapped to CelebAMask indices,
    long [1,1,256,256] (nearest resize) - or [1,1,H,W] with resize=False."""
    parsing = bisenet_logits(P, img_rgb)[0].argmax(0)
    remap = torch.tensor(LABEL_REMAP, dtype=parsing.dtype)
    mask = remap[parsing][None, None]
    if resize:  # torchvision resize(NEAREST) on an integer tensor == F.interpolate(mode='nearest')
        h, w = mask.shape[-2:]
        iy = (torch.arange(256) * (h / 256.0)).floor().long().clamp_(max=h - 1)
        ix = (torch.arange(256) * (w / 256.0)).floor().long().clamp_(max=w - 1)
        mask = mask[:, :, iy][:, :, :, ix]
    return mask


def bisenet_param_shapes(n_classes=19):
    """State-dict key -> shape of BiSeNet(n_classes) in the reference's key order."""
    S = {}

    def cbr(pre, cin, cout, ks):
        S[f"{pre}.conv.weight"] = (cout, cin, ks, ks)
        _bn_shapes(S, f"{pre}.bn", cout)

    r = "cp.resnet"
    S[f"{r}.conv1.weight"] = (64, 3, 7, 7)
    _bn_shapes(S, f"{r}.bn1", 64)
    inpl = 64
    for li, (planes, stride) in enumerate(((64, 1), (128, 2), (256, 2), (512, 2)), start=1):
        for j in range(2):
            b = f"{r}.layer{li}.{j}"
            S[f"{b}.conv1.weight"] = (planes, inpl, 3, 3)
            _bn_shapes(S, f"{b}.bn1", planes)
            S[f"{b}.conv2.weight"] = (planes, planes, 3, 3)
            _bn_shapes(S, f"{b}.bn2", planes)
            if j == 0 and (inpl != planes or stride != 1):
                S[f"{b}.downsample.0.weight"] = (planes, inpl, 1, 1)
                _bn_shapes(S, f"{b}.downsample.1", planes)
            inpl = planes
    for name, cin in (("cp.arm16", 256), ("cp.arm32", 512)):
        cbr(f"{name}.conv", cin, 128, 3)
        S[f"{name}.conv_atten.weight"] = (128, 128, 1, 1)
        _bn_shapes(S, f"{name}.bn_atten", 128)
    cbr("cp.conv_head32", 128, 128, 3)
    cbr("cp.conv_head16", 128, 128, 3)
    cbr("cp.conv_avg", 512, 128, 1)
    cbr("ffm.convblk", 256, 256, 1)
    S["ffm.conv1.weight"] = (64, 256, 1, 1)
    S["ffm.conv2.weight"] = (256, 64, 1, 1)
    for name, cin, mid in (("conv_out", 256, 256), ("conv_out16", 128, 64), ("conv_out32", 128, 64)):
        cbr(f"{name}.conv", cin, mid, 3)
        S[f"{name}.conv_out.weight"] = (n_classes, mid, 1, 1)
    return S
