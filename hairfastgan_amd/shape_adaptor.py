"""CtrlHair shape adaptor (mask generator) on the MI355X kernels - SURVEY.md section 8 row f4.

Host-side mirror of models/CtrlHair/shape_branch/model.py (`Generator`: two `MaskEncoder`s, two `MaskDecoder`s,
:69-186), the blocks of models/CtrlHair/my_torchlib/module.py (`Conv2dBlock` :64-131 with norm 'ln' = the MUNIT-style
`LayerNorm` :181-206, `LinearBlock` :16-61) and the two entry points HairFast calls, `get_hair_face_code` /
`get_new_shape` (models/CtrlHair/shape_branch/solver.py:248-262, call site models/Alignment.py:74-77): the hair shape
of the pose-rotated shape image is transplanted onto the face layout of the face image, as a label map.

Same class structure and state-dict keys as the reference (`hair_encoder.layers.N.conv.weight`, `...norm.gamma`,
`...out_layer.fc.weight`, `hair_decoder.layers.2N+1.conv.weight`, ...), configuration of
models/CtrlHair/shape_branch/config.py (hair_dim 16, g_norm 'ln', vae_hair_mode, pos_encoding_order 10).

How the layers run:
* encoder convs are 4x4 / stride 2 / zero pad 1: computed as a 3x3 / stride 1 convolution of the space-to-depth
  (pixel_unshuffle 2) padded input with re-laid-out weights - tap (2dy+py, 2dx+px) of input channel c becomes tap
  (1+dy, 1+dx) of channel 4c+2py+px, the other five taps are zero - on the library's conv kernels (the crop of the
  last row / column and the re-layout are torch glue; a handful of MFLOP);
* decoder convs (3x3, pad 1) after a nearest x2 up-sampling on the conv kernels (fp16 matrix cores where the shape
  allows), the LayerNorm + LeakyReLU as one kernel (hf_sample_layernorm_f32), Linear layers on hf_linear_f32;
* softmax + argmax of the 19 logits = argmax of the logits (torch glue, as the one-hot / positional-encoding input).
Inference only; the VAE resampling branch of the hair encoder is not used at test time (solver.py:254, testing=True).
"""
import math
import os

import torch
import torch.nn.functional as F
from torch import nn

from . import _marshal as M
from ._runtime import batch_invariant, conv_precision_scope, lib, require_gpu, stream
from .encoders._fused import FrozenPlanMixin, PreparedConv, conv, patches

# Trailing MaskDecoder convolutions (of eight) that run with exact fp32 products whatever the process-wide mode
# (HAIRFAST_SHAPE_EXACT_TAIL; tools/probes/shape_adaptor_flips.py measures the label-map flips against the reference golden)
EXACT_TAIL = int(os.environ.get("HAIRFAST_SHAPE_EXACT_TAIL", "0"))
HAIR_IDX = 13  # models/CtrlHair/global_value_utils.py:49-52 (PARSING_LABEL_LIST.index('hair'))


def generate_pos_embedding(img_size, order=10):  # model.py:18-30
    coord = torch.arange(img_size, dtype=torch.float64) / img_size
    xx, yy = torch.meshgrid(coord, coord, indexing="xy")
    bi = torch.stack([xx, yy], 0)[None]                                  # [1,2,S,S]
    nums = (2.0 ** torch.arange(order, dtype=torch.float64) * math.pi)[:, None, None, None]
    gamma = torch.cat([torch.sin(nums * bi), torch.cos(nums * bi)], 0)   # [2*order,2,S,S]
    return gamma.reshape(-1, img_size, img_size).float()


class LayerNorm(nn.Module):  # my_torchlib/module.py:181-206
    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.gamma = nn.Parameter(torch.rand(num_features))
        self.beta = nn.Parameter(torch.zeros(num_features))


class Conv2dBlock(nn.Module):  # my_torchlib/module.py:64-131 (pad_type 'zero', use_bias, norm in {'ln','none'}, lrelu 0.2 / none)
    def __init__(self, input_dim, output_dim, kernel_size, stride, padding=0, norm="none", activation="relu"):
        super().__init__()
        if norm not in ("ln", "none") or activation not in ("lrelu", "none"):
            raise NotImplementedError("the shape adaptor uses norm 'ln' / 'none' and LeakyReLU(0.2) / no activation")
        self.padding = padding
        self.conv = nn.Conv2d(input_dim, output_dim, kernel_size, stride, bias=True)
        self.norm = LayerNorm(output_dim) if norm == "ln" else None
        self.slope = 0.2 if activation == "lrelu" else 1.0
        self._plan = None

    def forward(self, x, const_planes=None):
        """const_planes [1, n, H, W]: trailing input channels that are the same for every sample and call (the positional
        encoding of a MaskEncoder's first layer): their share of the convolution is computed once per checkpoint and
        added as a per-pixel bias - the layer then only convolves the channels in `x`."""
        require_gpu(x)
        L, st = lib(), stream()
        k, s_ = self.conv.kernel_size[0], self.conv.stride[0]
        cout = self.conv.out_channels
        if self._plan is None:
            w = self.conv.weight.detach()
            if (k, s_, self.padding) == (4, 2, 1):  # -> 3x3 / stride 1 on the space-to-depth input (module docstring)
                def relayout(w4):
                    co, ci = w4.shape[:2]
                    w3 = w4.new_zeros(co, ci, 2, 2, 3, 3)
                    w3[:, :, :, :, 1:, 1:] = w4.reshape(co, ci, 2, 2, 2, 2).permute(0, 1, 3, 5, 2, 4)  # [co,ci,py,px,dy,dx]
                    return w3.reshape(co, ci * 4, 3, 3)
            elif (k, s_, self.padding) == (3, 1, 1):
                relayout = lambda w3: w3  # noqa: E731
            else:
                raise NotImplementedError("Conv2dBlock: 4x4 / stride 2 / pad 1 and 3x3 / stride 1 / pad 1")
            # The fp16 matrix-core kernel takes 16-channel K stages and whole 64-channel output tiles: input channels
            # are padded up to a multiple of 16 (zero planes / zero weights) and output channels to a multiple of 64
            # (zero filters; the extra planes are never read).  The general fp32 kernel ran these odd-shaped layers at
            # ~12 TFLOP/s (32 filters over 72 channels: 0.96 ms at batch 16).
            n_var = w.shape[1] - (0 if const_planes is None else const_planes.shape[1])
            pad_to = -(-cout // 64) * 64

            def prep(w_):
                w_ = relayout(w_)
                ci_pad = -(-w_.shape[1] // 16) * 16
                if ci_pad != w_.shape[1]:
                    w_ = torch.cat([w_, w_.new_zeros(w_.shape[0], ci_pad - w_.shape[1], 3, 3)], 1)
                if pad_to != cout:
                    w_ = torch.cat([w_, w_.new_zeros(pad_to - cout, *w_.shape[1:])], 0)
                return PreparedConv(M.conv_prepare(L, st, w_.contiguous()), 3)
            bias = self.conv.bias.detach()
            if pad_to != cout:
                bias = torch.cat([bias, bias.new_zeros(pad_to - cout)])
            # the conv-path weights are built on first use: the two deepest encoder layers (512 -> 1024 @ 8x8, 1024 -> 2048
            # @ 4x4) always take the patch-GEMM / GEMV branch below and never need their 75 + 302 MB space-to-depth form
            plan = {"w": None, "make_w": lambda: prep(w[:, :n_var]), "bias": bias.contiguous(), "pad_to": pad_to, "const": None}
            if const_planes is not None:  # conv of the constant planes (no bias: the variable part carries it)
                plan["const"] = self._conv(const_planes, prep(w[:, n_var:]), None, k)[:, :cout].contiguous()
            self._plan = plan
        p = self._plan
        if k == 4 and const_planes is None and x.shape[-1] <= 8 and x.shape[-2] <= 8:
            # Deep encoder layers (8x8 and 4x4 inputs, 512 / 1024 channels): a handful of output pixels, each a dot
            # product over cin*16 values - a GEMM [B*L, cin*16] x [cin*16, cout] whose cost is streaming the weights
            # once.  Patches by strided slices (glue, a few KB), the product on the 1x1 conv kernel; the space-to-depth form
            # would stream 2.25x the weights (its zero taps) through the general 3x3 path: 0.95 ms per layer at batch 16.
            b, _, h, w_ = x.shape
            cols = patches(x, 4, 2, 1, tap_major=False).flatten(2)                     # [B, cin*16, L], rows ordered like F.unfold's
            # batch-invariant plans (the default): the GEMM form folds the batch into its PIXEL axis - the library cannot plan
            # it for a canonical batch, its K split (the summation order of every sample, upstream of the label-map argmax)
            # would follow the real one - so the row-wise GEMV form is taken at any batch
            rows_planned = (1 if batch_invariant() else b) * cols.shape[2]
            if rows_planned > 32 and "gemm" not in p:  # the MFMA-GEMM form's weights: only when that branch runs
                w2 = self.conv.weight.detach().reshape(cout, -1, 1, 1)
                p["gemm"] = PreparedConv(M.conv_prepare(L, st, w2.contiguous()), 1)
            if rows_planned <= 32:
                # a single swap: up to 32 patch rows - the weight-streaming GEMV kernel reads the 34 / 134 MB of weights
                # once per 8 rows at HBM speed (the MFMA GEMM form below needs more rows to pay: 209 -> ~60 us)
                rows = cols.permute(0, 2, 1).reshape(-1, cols.shape[1])                # [B*L, cin*16]
                y = M.linear(L, st, rows, self.conv.weight.detach().reshape(cout, -1), self.conv.bias.detach(), 1.0)
                y = y.reshape(b, -1, cout).permute(0, 2, 1).reshape(b, cout, h // 2, w_ // 2).contiguous()
            else:
                cols = cols.permute(1, 0, 2).contiguous().unsqueeze(0)                     # [1, cin*16, B, L]
                # split-K GEMM on the fp16 matrix cores (csrc/gemm_h.hip): K = 8192 / 16384 over 64-256 patch columns is
                # weight streaming - 420-450 us on the fp32-MFMA 1x1 path
                y = conv(cols, p["gemm"], 1, 1, bias=self.conv.bias.detach())              # [1, cout, B, L]
                y = y[0].permute(1, 0, 2).reshape(b, cout, h // 2, w_ // 2).contiguous()
        else:
            if p["w"] is None:
                p["w"] = p.pop("make_w")()
            y = self._conv(x, p["w"], p["bias"], k)
        if p["const"] is not None:
            y = M.add_bcast(L, st, y, p["const"].reshape(-1)) if p["pad_to"] == cout else y[:, :cout] + p["const"]
        if self.norm is not None:
            return M.sample_layernorm(L, st, y, self.norm.gamma.detach(), self.norm.beta.detach(), self.norm.eps, self.slope,
                                      channels=cout)
        y = y if y.shape[1] == cout else y[:, :cout].contiguous()
        return y if self.slope == 1.0 else F.leaky_relu(y, self.slope)

    @staticmethod
    def _conv(x, w, bias, k):
        h, w_ = x.shape[-2:]
        if k == 4:
            x = F.pixel_unshuffle(F.pad(x, (1, 1, 1, 1)), 2)
        if x.shape[1] != w.cin:  # zero planes up to the prepared weights' (padded) input-channel count
            x = F.pad(x, (0, 0, 0, 0, 0, w.cin - x.shape[1]))
        y = conv(x, w, 3, 1, bias=bias)
        return y[:, :, :h // 2, :w_ // 2].contiguous() if k == 4 else y


class LinearBlock(nn.Module):  # my_torchlib/module.py:16-61 (norm 'none', activation 'none')
    def __init__(self, input_dim, output_dim):
        super().__init__()
        self.fc = nn.Linear(input_dim, output_dim, bias=True)

    def forward(self, x):
        require_gpu(x)
        return M.linear(lib(), stream(), x, self.fc.weight.detach(), self.fc.bias.detach(), 1.0)


class MaskEncoder(FrozenPlanMixin, nn.Module):  # model.py:69-115
    def __init__(self, input_channel, output_dim, layer_num=7, input_size=256, vae_mode=False, pos_encoding_order=10,
                 hidden_in_channel=32):
        super().__init__()
        self.vae_mode = vae_mode
        layers, in_channel = [], input_channel + pos_encoding_order * 4
        for cur in range(layer_num):
            out_channel = min(2048, 2 ** cur * hidden_in_channel)
            layers.append(Conv2dBlock(in_channel, out_channel, 4, 2, padding=1, norm="ln", activation="lrelu"))
            in_channel = out_channel
        self.layers = nn.Sequential(*layers)
        fc_in_dim = (input_size // 2 ** layer_num) ** 2 * out_channel
        self.out_layer = LinearBlock(fc_in_dim, output_dim)
        if vae_mode:
            self.std_out_layer = LinearBlock(fc_in_dim, output_dim)  # training-time resampling: unused here
        self.register_buffer("input_embedding", generate_pos_embedding(input_size, pos_encoding_order)[None], persistent=False)

    def forward(self, input_mask):
        """[B, input_channel, 256, 256] one-hot planes -> mean code [B, output_dim] (`out_mean`, model.py:103-108)."""
        # x = cat([input_mask, positional encoding]) (model.py:101): the encoding's share of the first conv is a constant
        feature = self.layers[0](input_mask, const_planes=self.input_embedding)
        for layer in list(self.layers)[1:]:
            feature = layer(feature)
        return self.out_layer(feature.flatten(1))


class MaskDecoder(FrozenPlanMixin, nn.Module):  # model.py:118-146
    def __init__(self, input_dim, output_channel, layer_num=7, output_size=256):
        super().__init__()
        self.in_channel = min(32 * 2 ** layer_num, 2048)
        self.input_size = output_size // 2 ** layer_num
        self.in_layer = LinearBlock(input_dim, self.in_channel * self.input_size ** 2)
        layers, in_channel = [], self.in_channel
        for cur in range(layer_num):
            out_channel = min(32 * 2 ** (layer_num - 1 - cur), 2048)
            layers.append(nn.Upsample(scale_factor=2, mode="nearest"))  # a parameter-free slot (keeps the reference's indices)
            layers.append(Conv2dBlock(in_channel, out_channel, 3, 1, padding=1, norm="ln", activation="lrelu"))
            in_channel = out_channel
        self.layers = nn.Sequential(*layers)
        self.out_layer = Conv2dBlock(in_channel, output_channel, 3, 1, padding=1, norm="none", activation="none")

    def forward(self, input_vector):
        L, st = lib(), stream()
        x = self.in_layer(input_vector).reshape(-1, self.in_channel, self.input_size, self.input_size)
        # EXACT_TAIL: the last n of the decoder's eight convolutions (seven blocks + the logit layer) with exact fp32 products
        # (fp32 MFMA) whatever the process-wide mode: what feeds the label-map argmax directly
        convs = [m for m in self.layers if not isinstance(m, nn.Upsample)]
        exact_from = len(convs) + 1 - EXACT_TAIL
        seen = 0
        for layer in self.layers:
            if isinstance(layer, nn.Upsample):
                x = M.upsample_nearest(L, st, x, 2 * x.shape[2], 2 * x.shape[3])
            else:
                with conv_precision_scope("f32" if seen >= exact_from else None):
                    x = layer(x)
                seen += 1
        with conv_precision_scope("f32" if EXACT_TAIL >= 1 else None):
            return self.out_layer(x)


class MaskGenerator(nn.Module):
    """models/CtrlHair/shape_branch/model.py:149-186 `Generator` (cfg: hair_dim 16, vae_hair_mode True, g_norm 'ln')."""

    def __init__(self, hair_dim=16, pos_encoding_order=10):
        super().__init__()
        self.hair_encoder = MaskEncoder(1, hair_dim, vae_mode=True, pos_encoding_order=pos_encoding_order)
        self.face_encoder = MaskEncoder(18, 1024, vae_mode=False, pos_encoding_order=pos_encoding_order)
        self.hair_decoder = MaskDecoder(1024 + hair_dim, output_channel=1)
        self.face_decoder = MaskDecoder(1024, output_channel=18)

    def forward_hair_encoder(self, hair, testing=True):
        if not testing:
            raise NotImplementedError("training-time VAE resampling (model.py:110-113): inference only")
        return self.hair_encoder(hair)

    def forward_face_encoder(self, face):
        return self.face_encoder(face)

    def forward_decode_by_code(self, hair_code, face_code):
        """-> class logits [B,19,256,256] in label order (model.py:171-186 before the softmax: its argmax is theirs)."""
        hair_logit = self.hair_decoder(torch.cat([face_code, hair_code], dim=1))
        face_logit = self.face_decoder(face_code)
        return torch.cat([face_logit[:, :HAIR_IDX], hair_logit, face_logit[:, HAIR_IDX:]], dim=1)


def mask_label_to_one_hot(img):  # shape_util.py:6-14: [B,1,H,W] labels (255 = none) -> [B,19,H,W]
    img = torch.where(img == 255, torch.full_like(img, 19), img).long()
    b, _, h, w = img.shape
    return torch.zeros(b, 20, h, w, device=img.device).scatter_(1, img, 1.0)[:, :-1]


def split_hair_face(mask):  # shape_util.py:23-26
    return mask[:, HAIR_IDX:HAIR_IDX + 1], torch.cat([mask[:, :HAIR_IDX], mask[:, HAIR_IDX + 1:]], dim=1)  # (a list index would upload an index tensor: illegal in a hipGraph capture)


@torch.inference_mode()
def get_hair_face_code(mask_generator, mask):  # solver.py:248-256; mask [H,W] (or [B,H,W]) label map, 256 x 256
    mask_batch = mask[None, None] if mask.ndim == 2 else mask[:, None]
    if tuple(mask_batch.shape[-2:]) != (256, 256):
        mask_batch = F.interpolate(mask_batch.float(), size=(256, 256), mode="nearest")
    hair, face = split_hair_face(mask_label_to_one_hot(mask_batch.long()))
    return mask_generator.forward_face_encoder(face.contiguous()), mask_generator.forward_hair_encoder(hair.contiguous(), testing=True)


@torch.inference_mode()
def get_new_shape(mask_generator, face_code, new_hair_code):  # solver.py:259-262 -> label map [256,256] (first of the batch)
    return new_shape_batch(mask_generator, face_code, new_hair_code)[0]


@torch.inference_mode()
def new_shape_batch(mask_generator, face_code, new_hair_code):
    """get_new_shape for a batch of codes: [B,256,256] label maps (softmax is monotonic; its maximum is never 0, so the
    reference's `mask[max == 0] = 255`, shape_util.py:19, never fires)."""
    return mask_generator.forward_decode_by_code(new_hair_code, face_code).argmax(dim=1)


@torch.inference_mode()
def adapt_shape(mask_generator, mask_target_pose, mask_hair_source):
    """The shape-adaptor stage of Alignment.shape_module (Alignment.py:74-77) for B pairs at once: face code of the
    face image's mask, hair code of the rotated shape image's mask -> target label maps, long [B,1,256,256].
    (The reference also encodes the two unused codes - hair of the face image, face of the shape image; skipped.)"""
    b = mask_target_pose.shape[0]
    if tuple(mask_target_pose.shape[-2:]) != (256, 256) or tuple(mask_hair_source.shape[-2:]) != (256, 256):
        raise NotImplementedError("256 x 256 label maps (what get_segmentation returns)")
    _, face = split_hair_face(mask_label_to_one_hot(mask_target_pose.reshape(b, 1, 256, 256)))
    hair, _ = split_hair_face(mask_label_to_one_hot(mask_hair_source.reshape(b, 1, 256, 256)))
    face_code = mask_generator.forward_face_encoder(face.contiguous())
    hair_code = mask_generator.forward_hair_encoder(hair.contiguous(), testing=True)
    return new_shape_batch(mask_generator, face_code, hair_code)[:, None]
