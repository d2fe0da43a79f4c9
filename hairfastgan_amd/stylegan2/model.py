"""StyleGAN2 generator on the MI355X kernels - host-side mirror of the reference's
models/stylegan2/model.py (all file:line below refer to that file).

Same class names, constructor arguments, parameter/buffer names (state-dict
compatible: 171 entries for Generator(1024, 512, 8, 2)) and `forward` signatures
/ return conventions, so reference-style callers

    net.generator([latent], input_is_latent=True, return_latents=False,
                  start_layer=4, end_layer=8, layer_in=F)        # -> (tensor, tensor|None)

work unchanged.  What differs is the execution: a StyledConv is 3 kernel launches
(modulation, demodulation coefficients, fused fp32-MFMA conv + noise + bias +
leaky-ReLU; 5 for the upsampling variant), a ToRGB 2, instead of the reference's
6-8 ATen calls on materialised per-sample weights (:238-279).

Forward / inference only: parameters are frozen in HairFast (models/Net.py:44-46)
and every stage runs under torch.inference_mode().
"""
import math
import os
import random

import torch
from torch import nn
from torch.nn import functional as F

from .. import _marshal as M
from .._runtime import conv_precision, lib, reference_rng_walk, require_gpu, run_guarded, stream
from .op import FusedLeakyReLU, fused_leaky_relu, upfirdn2d


# Smallest input height of an upsampling StyledConv that runs as one fused kernel (transposed conv + blur +
# noise + bias + lrelu, csrc/convh.hip FUSE) instead of conv -> (2h+1)^2 intermediate -> blur pass; the fused
# form recomputes a 1-position halo per 16x32 tile (+22 % matrix work) and saves 8 bytes of HBM traffic per
# intermediate element, which pays on the bandwidth-bound high-resolution layers (batch 8, tools/probes/fuse_layers.py:
# 512->1024: 1100 -> 639 us, 256->512: 604 -> 441, 128->256: 456 -> 425, 64->128: 351 -> 408 - hence 128).
FUSE_BLUR_MIN_H = int(os.environ.get("HAIRFAST_FUSE_BLUR_MIN_H", "128"))


class PixelNorm(nn.Module):  # :16-21 (mapping network only; not on the HairFast hot path)
    def forward(self, input):
        require_gpu(input)
        return M.pixel_norm(lib(), stream(), input)


def make_kernel(k):  # :24-32
    k = torch.tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = k[None, :] * k[:, None]
    return k / k.sum()


class Upsample(nn.Module):  # :35-53
    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        self.register_buffer("kernel", make_kernel(kernel) * (factor ** 2))
        p = self.kernel.shape[0] - factor
        self.pad = ((p + 1) // 2 + factor - 1, p // 2)

    def forward(self, input):
        return upfirdn2d(input, self.kernel, up=self.factor, down=1, pad=self.pad)


class Downsample(nn.Module):  # :56-74
    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        self.register_buffer("kernel", make_kernel(kernel))
        p = self.kernel.shape[0] - factor
        self.pad = ((p + 1) // 2, p // 2)

    def forward(self, input):
        return upfirdn2d(input, self.kernel, up=1, down=self.factor, pad=self.pad)


class Blur(nn.Module):  # :77-93
    def __init__(self, kernel, pad, upsample_factor=1):
        super().__init__()
        kernel = make_kernel(kernel)
        if upsample_factor > 1:
            kernel = kernel * (upsample_factor ** 2)
        self.register_buffer("kernel", kernel)
        self.pad = pad

    def forward(self, input):
        return upfirdn2d(input, self.kernel, pad=self.pad)


class EqualLinear(nn.Module):  # :134-168
    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1, activation=None):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.zeros(out_dim).fill_(bias_init)) if bias else None
        self.activation = activation
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul

    def forward(self, input):
        # Only the z->w mapping network reaches this generic form (weight-streaming GEMV batch, one
        # launch of hf_equal_linear_f32 per layer incl. bias*lr_mul and the fused leaky ReLU; not on
        # HairFast's path: input_is_latent=True everywhere); the 26 modulation linears go through
        # hf_modulation_f32 / hf_style_batch_f32 inside ModulatedConv2d.
        require_gpu(input)
        return M.equal_linear(lib(), stream(), input, self.weight.detach(),
                              None if self.bias is None else self.bias.detach(), self.lr_mul, bool(self.activation))


class ModulatedConv2d(nn.Module):  # :183-279
    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True, upsample=False,
                 downsample=False, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        if downsample:
            raise NotImplementedError("downsample=True is Discriminator-only in the reference (out of scope)")
        if kernel_size not in (1, 3):
            raise NotImplementedError("kernel_size must be 1 (ToRGB) or 3 (StyledConv)")
        if upsample and kernel_size != 3:
            raise NotImplementedError("upsample needs kernel_size 3")
        if kernel_size == 1 and (out_channel != 3 or demodulate):
            raise NotImplementedError("the 1x1 modulated conv exists only as ToRGB (3 outputs, no demodulation)")
        self.eps = 1e-8
        self.kernel_size = kernel_size
        self.in_channel = in_channel
        self.out_channel = out_channel
        self.upsample = upsample
        self.downsample = downsample
        if upsample:
            factor = 2
            p = (len(blur_kernel) - factor) - (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2 + factor - 1, p // 2 + 1), upsample_factor=factor)
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.padding = kernel_size // 2
        self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size))
        self.modulation = EqualLinear(style_dim, in_channel, bias_init=1)
        self.demodulate = demodulate
        self._prep = None  # (key, wt [k*k,cin,cout], wsq [cout,cin]) - derived, not in the state dict
        self._prep_f16 = None  # (key, wt_hi, wt_lo) fp16 split of wt for the fp16 matrix-core path
        self._coeffs = None  # ((style ptr, shape, stride), (wt, s, d)) set for the duration of one Generator.forward
        # a (re)load - also through a parent container - drops the derived tensors: the cache key
        # (data_ptr, version) cannot see an in-place copy into an inference tensor
        self.register_load_state_dict_post_hook(lambda m, _ik: m.invalidate())

    def invalidate(self):
        self._prep = self._prep_f16 = self._coeffs = None

    def __repr__(self):
        return (f"{self.__class__.__name__}({self.in_channel}, {self.out_channel}, {self.kernel_size}, "
                f"upsample={self.upsample}, downsample={self.downsample})")

    # -- derived, frozen-parameter state --------------------------------------------------
    def prepared(self):
        """Batch-shared re-layout of the weight: wt[tap][ci][co] = scale*W, wsq[co][ci] = sum_tap wt^2."""
        w = self.weight
        # inference tensors (module moved/loaded under torch.inference_mode()) carry no version counter
        key = (w.data_ptr(), None if w.is_inference() else w._version, w.device)
        if self._prep is None or self._prep[0] != key:
            require_gpu(w)
            wt, wsq = M.prepare_weights(lib(), stream(), w.detach())
            self._prep = (key, wt, wsq)
        return self._prep[1], self._prep[2]

    def prepared_f16(self):
        """(hi, lo) fp16 split of the prepared weight in csrc/convh.hip's layout (cached)."""
        wt, _ = self.prepared()
        key = self._prep[0]
        if self._prep_f16 is None or self._prep_f16[0] != key:
            hi, lo = M.split_weights_f16(lib(), stream(), wt)
            self._prep_f16 = (key, hi, lo)
        return self._prep_f16[1], self._prep_f16[2]

    def prepared_small(self):
        """(hi, lo) of the prepared weight as the rows of ONE 1x1 GEMM, [cin][tap*cout + co] (M.split_weights_small; cached)."""
        wt, _ = self.prepared()
        key = self._prep[0]
        cached = self.__dict__.get("_prep_small")
        if cached is None or cached[0] != key:
            cached = self.__dict__["_prep_small"] = (key, M.split_weights_small(lib(), stream(), wt))
        return cached[1]

    def conv_same_res(self, input, wt, s, d, noise, noise_w, bias, alpha=0.2, scale=math.sqrt(2), rgb=None):
        """3x3 same-resolution modulated conv (+ fused noise/bias/lrelu epilogue) on the matrix
        cores the process-wide mode selects (_runtime.conv_precision).  rgb: see fuses_torgb."""
        mode = conv_precision()
        b_, cin, h, w = input.shape
        if mode != "f32" and rgb is None and M.modconv3x3_small_supported(cin, self.out_channel, h, w, b_):
            # small planes (the 4^2 - 32^2 tower at small batch): nine taps as one GEMM over the whole chip + a combine pass
            return M.modconv3x3_small(lib(), stream(), input, self.prepared_small(), 3 if mode == "f16x3" else 1, s, d, noise,
                                      noise_w, bias, self.out_channel, alpha, scale)
        if mode != "f32" and M.modconv3x3_f16_supported(cin, self.out_channel, h, w, batch=b_):
            hi, lo = self.prepared_f16()
            return M.modconv3x3_f16(lib(), stream(), input, hi, lo, 3 if mode == "f16x3" else 1, s, d, noise,
                                    noise_w, bias, alpha, scale, rgb=rgb)
        assert rgb is None
        return M.modconv3x3(lib(), stream(), input, wt, s, d, noise, noise_w, bias, alpha, scale)

    def fuses_torgb(self, input):
        """True when the layer's ToRGB can be computed in this conv's epilogue (fp16-core modes; a wave sums the
        64 / 32 output channels it holds, ToRGB.finish adds the slabs)."""
        b_, cin, h, w = input.shape
        # the same predicates conv_same_res dispatches on: the fp16-core kernel must be the one that runs (batch-dependent)
        return (conv_precision() != "f32" and not self.upsample and self.kernel_size == 3
                and M.torgb_fusable(cin, self.out_channel, h, w)
                and M.modconv3x3_f16_supported(cin, self.out_channel, h, w, batch=b_))

    def blur_factors(self):
        """1-D factors of the module's blur kernel when it is separable (it is: make_kernel of a 1-D list,
        :24-32), else None; read once per kernel buffer (a host round trip)."""
        k = self.blur.kernel
        key = (k.data_ptr(), k._version if not k.is_inference() else None)
        cached = self.__dict__.get("_blur_fac")
        if cached is None or cached[0] != key:
            if torch.cuda.is_current_stream_capturing():
                return None if cached is None else cached[1]
            cached = self.__dict__["_blur_fac"] = (key, M.blur_factors(k))
        return cached[1]

    def conv_up(self, input, wt, s, d, noise, noise_w, bias, alpha=0.2, scale=math.sqrt(2), split_for=None):
        """Transposed 3x3 conv + blur (+ fused noise/bias/lrelu), matrix cores per the mode.
        split_for=(key, s_next): the blur pass writes a SplitActivation for the next conv.
        Large planes in the fp16-core modes run the ONE-kernel form (hf_modconv3x3_up_blur_f16_f32: no (2h+1)^2
        intermediate); FUSE_BLUR_MIN_H is the smallest input height that takes it."""
        mode = conv_precision()
        _, cin, h, w = input.shape
        if (mode in ("f16x3", "f16") and h >= FUSE_BLUR_MIN_H and noise is not None and bias is not None and 0.0 <= alpha <= 1.0
                and M.modconv3x3_up_fused_supported(cin, self.out_channel, h, w)):
            fac = self.blur_factors()
            if fac is not None:
                hi, lo = self.prepared_f16()
                return M.modconv3x3_up_fused(lib(), stream(), input, hi, lo, s, d, fac, noise, noise_w, bias, alpha, scale,
                                             split_for=None if split_for is None else split_for[1],
                                             nterms=3 if mode == "f16x3" else 1)
        f16 = small = None
        if (mode != "f32" and not isinstance(input, M.SplitActivation)
                and M.modconv3x3_small_supported(cin, self.out_channel, h, w, input.shape[0], upsample=True)):
            small = (self.prepared_small(), 3 if mode == "f16x3" else 1)  # small planes: the nine taps as one GEMM
        elif mode != "f32" and M.modconv3x3_up_f16_supported(cin, self.out_channel, h, w, batch=input.shape[0]):
            hi, lo = self.prepared_f16()
            f16 = (hi, lo, 3 if mode == "f16x3" else 1)
        return M.modconv3x3_up(lib(), stream(), input, wt, s, d, self.blur.kernel, noise, noise_w, bias, alpha, scale,
                               f16=f16, split_for=split_for, small=small)

    def style_coefficients(self, style):
        """s[b,ci] (EqualLinear :241) and d[b,co] (:244-246; None when demodulate=False).  With
        demodulation the pair is returned range-normalised (s*2^-e, d*2^e per sample; exact)."""
        pre = self._coeffs
        if pre is not None and pre[0] == (style.data_ptr(), tuple(style.shape), tuple(style.stride())):
            return pre[1]  # computed for this very row of W+ by Generator.forward's batched launch
        wt, wsq = self.prepared()
        s = M.modulation(lib(), stream(), style, self.modulation.weight.detach(), self.modulation.bias.detach())
        d = None
        if self.demodulate:
            d = M.demod(lib(), stream(), s, wsq)
            # exact power-of-two rescaling s*2^-e, d*2^e (the conv is invariant): keeps s*x inside the
            # range of the fp16 (hi, lo) operand split whatever the trained style's magnitude
            M.style_normalize(lib(), stream(), s, d)
        return wt, s, d

    def forward(self, input, style):
        require_gpu(input, style)
        wt, s, d = self.style_coefficients(style)
        if self.kernel_size == 1:
            return M.torgb(lib(), stream(), input, wt, s, None, None, None)
        if self.upsample:
            if tuple(self.blur.pad) != (1, 1) or tuple(self.blur.kernel.shape) != (4, 4):
                raise NotImplementedError("fused upsampling path expects the [1,3,3,1] blur with pad (1,1)")
            return self.conv_up(input, wt, s, d, None, None, None)
        return self.conv_same_res(input, wt, s, d, None, None, None)


class NoiseInjection(nn.Module):  # :282-293
    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1))

    def forward(self, image, noise=None):
        require_gpu(image)
        if noise is None:
            batch, _, height, width = image.shape
            noise = image.new_empty(batch, 1, height, width).normal_()
        return M.noise_bias_act(lib(), stream(), image, noise, self.weight.detach(), None, 1.0, 1.0)


class ConstantInput(nn.Module):  # :296-306
    def __init__(self, channel, size=4):
        super().__init__()
        self.input = nn.Parameter(torch.randn(1, channel, size, size))

    def forward(self, input):
        return self.input.repeat(input.shape[0], 1, 1, 1)


class StyledConv(nn.Module):  # :309-343
    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False,
                 blur_kernel=[1, 3, 3, 1], demodulate=True):
        super().__init__()
        self.conv = ModulatedConv2d(in_channel, out_channel, kernel_size, style_dim, upsample=upsample,
                                    blur_kernel=blur_kernel, demodulate=demodulate)
        self.noise = NoiseInjection()
        self.activate = FusedLeakyReLU(out_channel)

    def forward(self, input, style, noise=None):
        """conv -> +noise -> bias -> leaky-ReLU*sqrt2 with the last three fused into the
        producing kernel's epilogue (same-res) or into the blur pass / the one-kernel upsampling form."""
        return self._run(input, style, noise, None)

    def forward_rgb(self, input, style, noise, rgb):
        """forward() that ALSO returns ToRGB's raw 1x1 modulated conv of its output, computed in the same
        epilogue (only when conv.fuses_torgb(input)): (out, raw [B,3,H,W]).  rgb = ToRGB.coefficients(style) of the
        layer's ToRGB; the caller hands `raw` to that ToRGB's finish() instead of letting it re-read the feature
        map.  An explicit second return value - nothing is attached to tensors."""
        return self._run(input, style, noise, rgb)

    def _run(self, input, style, noise, rgb):
        require_gpu(input, style, noise)
        conv = self.conv
        wt, s, d = conv.style_coefficients(style)
        b, _, h, w = input.shape
        oh, ow = (2 * h, 2 * w) if conv.upsample else (h, w)
        if noise is None:  # fresh N(0,1) per call, drawn from torch's device RNG like :289-291
            noise = input.new_empty(b, 1, oh, ow).normal_()
        act = self.activate
        if conv.upsample:
            assert rgb is None
            return conv.conv_up(input, wt, s, d, noise, self.noise.weight.detach(), act.bias.detach(),
                                act.negative_slope, act.scale)
        if rgb is None:
            return conv.conv_same_res(input, wt, s, d, noise, self.noise.weight.detach(), act.bias.detach(),
                                      act.negative_slope, act.scale)
        rgb_wt, rgb_s = rgb
        return conv.conv_same_res(input, wt, s, d, noise, self.noise.weight.detach(), act.bias.detach(),
                                  act.negative_slope, act.scale, rgb=(rgb_wt, rgb_s))

    # -- producer -> consumer hand-over without an fp32 round trip (Generator's fast path) --------
    def forward_split(self, input, style, noise, s_next, coeffs=None):
        """The upsampling StyledConv whose only consumer is the next same-resolution StyledConv:
        returns s_next * output already split into fp16 (hi, lo) and K-blocked (M.SplitActivation)
        instead of the fp32 tensor; same arithmetic as forward()."""
        pre = isinstance(input, M.SplitActivation)  # its own input may come pre-split from the layer below
        require_gpu(None if pre else input, style, noise)
        conv = self.conv
        assert conv.upsample
        wt, s, d = conv.style_coefficients(style) if coeffs is None else coeffs  # coeffs: computed by the caller already
        b, _, h, w = input.shape
        if noise is None:
            noise = style.new_empty(b, 1, 2 * h, 2 * w).normal_()
        act = self.activate
        return conv.conv_up(input, wt, s, d, noise, self.noise.weight.detach(), act.bias.detach(),
                            act.negative_slope, act.scale, split_for=(None, s_next, conv_precision() == "f16x3"))

    def forward_image(self, split, coeffs, noise, rgb, to_rgb, skip):
        """The generator's last StyledConv + its ToRGB in ONE launch (M.image_fusable shapes): the image, or None when the
        library declines (the caller then takes forward_from_split + ToRGB.finish - the same bits)."""
        conv, act = self.conv, self.activate
        _, s, d = coeffs
        b, _, h, w = split.shape
        if noise is None:
            noise = split.hi.new_empty(b, 1, h, w, dtype=torch.float32).normal_()
        require_gpu(noise, skip)
        hi, lo = conv.prepared_f16()
        nterms = 3 if conv_precision() == "f16x3" else 1
        return M.modconv3x3_f16_pre_image(lib(), stream(), split, hi, lo, nterms, d, noise, self.noise.weight.detach(),
                                          act.bias.detach(), rgb, to_rgb.bias.detach(), skip, to_rgb._skip_kernel(skip),
                                          act.negative_slope, act.scale)

    def forward_from_split(self, split, coeffs, noise=None, rgb=None, want_out=True, split_for=None):
        """Same-resolution StyledConv on a SplitActivation produced for it (coeffs = this layer's
        conv.style_coefficients(style), whose s went into the split).  Returns (out, raw, next_split):
        out is what forward() returns, or None with want_out=False (nobody reads the fp32 activation);
        raw (rgb = ToRGB.coefficients of the layer's ToRGB) is that ToRGB's 1x1 conv from the epilogue, else None;
        next_split (split_for = the next layer's modulation) is the SplitActivation for the transposed conv
        above, else None."""
        conv = self.conv
        assert not conv.upsample
        _, s, d = coeffs
        b, _, h, w = split.shape
        if noise is None:
            noise = split.hi.new_empty(b, 1, h, w, dtype=torch.float32).normal_()
        require_gpu(noise)
        act = self.activate
        hi, lo = conv.prepared_f16()
        nterms = 3 if conv_precision() == "f16x3" else 1
        res = M.modconv3x3_f16_pre(lib(), stream(), split, hi, lo, nterms, d, noise, self.noise.weight.detach(),
                                   act.bias.detach(), act.negative_slope, act.scale, rgb=rgb, want_out=want_out,
                                   split_for=split_for)
        res = list(res) if isinstance(res, tuple) else [res]
        out = res.pop(0)
        raw = res.pop(0) if rgb is not None else None
        nxt = res.pop(0) if split_for is not None else None
        return out, raw, nxt


def _observed(*modules):
    """True when someone registered hooks on these modules or their children: the fused fast
    paths then step aside so that every module is called and sees / returns plain tensors."""
    for m in modules:
        for sub in m.modules():
            if sub._forward_hooks or sub._forward_pre_hooks:
                return True
    return False


class ToRGB(nn.Module):  # :346-365
    def __init__(self, in_channel, style_dim, upsample=True, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        if upsample:
            self.upsample = Upsample(blur_kernel)
        self.conv = ModulatedConv2d(in_channel, 3, 1, style_dim, demodulate=False)
        self.bias = nn.Parameter(torch.zeros(1, 3, 1, 1))

    def _skip_kernel(self, skip):
        if skip is None:
            return None
        kern = self.upsample.kernel
        if tuple(kern.shape) != (4, 4) or self.upsample.factor != 2 or tuple(self.upsample.pad) != (2, 1):
            raise NotImplementedError("fused skip path expects the [1,3,3,1] x2 upsampler")
        return kern

    def forward(self, input, style, skip=None):
        require_gpu(input, style, skip)
        wt, s, _ = self.conv.style_coefficients(style)
        return M.torgb(lib(), stream(), input, wt, s, self.bias.detach(), skip, self._skip_kernel(skip))

    def coefficients(self, style):
        """(wt [1,cin,3], s [B,cin]) for the producer that fuses the 1x1 conv (StyledConv.forward_rgb)."""
        wt, s, _ = self.conv.style_coefficients(style)
        return wt, s

    def finish(self, raw, skip=None):
        """forward() when the producing StyledConv already computed the raw 1x1 modulated conv in its epilogue
        (StyledConv.forward_rgb / forward_from_split): bias + upsampled skip on top of `raw` - the same kernel
        with a 3-channel input and identity weights."""
        require_gpu(raw, skip)
        slabs = raw.shape[1] // 3  # partial sums over 64 (32) output channels each: added here, in a fixed order
        eye = getattr(self, "_eye", None)
        if eye is None or eye.device != raw.device or eye.shape[1] != 3 * slabs:
            eye = self._eye = torch.eye(3, device=raw.device, dtype=raw.dtype).repeat(slabs, 1).reshape(1, 3 * slabs, 3)
        return M.torgb(lib(), stream(), raw, eye, None, self.bias.detach(), skip, self._skip_kernel(skip))


class Generator(nn.Module):  # :368-565
    def __init__(self, size, style_dim, n_mlp, channel_multiplier=2, blur_kernel=[1, 3, 3, 1], lr_mlp=0.01):
        super().__init__()
        self.size = size
        self.style_dim = style_dim
        layers = [PixelNorm()]
        for _ in range(n_mlp):
            layers.append(EqualLinear(style_dim, style_dim, lr_mul=lr_mlp, activation="fused_lrelu"))
        self.style = nn.Sequential(*layers)
        self.channels = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * channel_multiplier,
                         128: 128 * channel_multiplier, 256: 64 * channel_multiplier,
                         512: 32 * channel_multiplier, 1024: 16 * channel_multiplier}
        self.input = ConstantInput(self.channels[4])
        self.conv1 = StyledConv(self.channels[4], self.channels[4], 3, style_dim, blur_kernel=blur_kernel)
        self.to_rgb1 = ToRGB(self.channels[4], style_dim, upsample=False)
        self.log_size = int(math.log(size, 2))
        self.num_layers = (self.log_size - 2) * 2 + 1
        self.convs = nn.ModuleList()
        self.upsamples = nn.ModuleList()
        self.to_rgbs = nn.ModuleList()
        self.noises = nn.Module()
        for layer_idx in range(self.num_layers):
            res = (layer_idx + 5) // 2
            self.noises.register_buffer(f"noise_{layer_idx}", torch.randn(1, 1, 2 ** res, 2 ** res))
        in_channel = self.channels[4]
        for i in range(3, self.log_size + 1):
            out_channel = self.channels[2 ** i]
            self.convs.append(StyledConv(in_channel, out_channel, 3, style_dim, upsample=True,
                                         blur_kernel=blur_kernel))
            self.convs.append(StyledConv(out_channel, out_channel, 3, style_dim, blur_kernel=blur_kernel))
            self.to_rgbs.append(ToRGB(out_channel, style_dim))
            in_channel = out_channel
        self.n_latent = self.log_size * 2 - 2
        self.register_load_state_dict_post_hook(lambda m, _ik: m.__dict__.pop("_style_jobs", None) and None)

    def make_noise(self):  # :455-464
        device = self.input.input.device
        noises = [torch.randn(1, 1, 2 ** 2, 2 ** 2, device=device)]
        for i in range(3, self.log_size + 1):
            for _ in range(2):
                noises.append(torch.randn(1, 1, 2 ** i, 2 ** i, device=device))
        return noises

    def mean_latent(self, n_latent):  # :466-472
        latent_in = torch.randn(n_latent, self.style_dim, device=self.input.input.device)
        return self.style(latent_in).mean(0, keepdim=True)

    def get_latent(self, input):
        return self.style(input)

    def forward(self, styles, return_latents=False, inject_index=None, truncation=1, truncation_latent=None,
                input_is_latent=False, noise=None, randomize_noise=True, layer_in=None, skip=None,
                start_layer=0, end_layer=8, return_rgb=False):
        """Layer-range executor with the reference's semantics (:477-565): returns
        (image, latent|None) after the last block, (feature, skip) on early exit."""
        if not input_is_latent:
            styles = [self.style(s) for s in styles]
        if noise is None:
            if randomize_noise:
                noise = [None] * self.num_layers
            else:
                noise = [getattr(self.noises, f"noise_{i}") for i in range(self.num_layers)]
        if truncation < 1:
            styles = [truncation_latent + truncation * (s - truncation_latent) for s in styles]
        if len(styles) < 2:
            inject_index = self.n_latent
            latent = styles[0].unsqueeze(1).repeat(1, inject_index, 1) if styles[0].ndim < 3 else styles[0]
        else:
            if inject_index is None:
                inject_index = random.randint(1, self.n_latent - 1)
            latent = torch.cat([styles[0].unsqueeze(1).repeat(1, inject_index, 1),
                                styles[1].unsqueeze(1).repeat(1, self.n_latent - inject_index, 1)], 1)

        if layer_in is None or tuple(layer_in.shape[-2:]) == (2 ** (start_layer + 1),) * 2:  # standard pyramid sizes only
            noise = self._draw_noise(noise, latent, start_layer, end_layer)

        def run():  # HAIRFAST_CONV_PRECISION=auto: repeated on the fp32 kernels (same latent, same noise) if the split clamped
            styled = self._batch_styles(latent)
            try:
                return self._run_layers(latent, noise, layer_in, skip, start_layer, end_layer, return_latents)
            finally:
                for conv in styled:
                    conv._coeffs = None

        return run_guarded(run)

    def _draw_noise(self, noise, latent, start_layer, end_layer):
        """randomize_noise: the fresh N(0,1) maps of every layer that will run, from ONE normal_() launch on torch's
        device RNG (the reference draws one tensor per layer inside NoiseInjection, :289-291 - 17 launches per forward;
        same distribution, same seed -> same images, but a different walk through the Philox stream: bit-level
        noise parity with a CUDA run does not exist either way).  Stand-alone layers still draw their own."""
        if any(n is not None for n in noise) or not latent.is_cuda or reference_rng_walk():
            return noise  # HAIRFAST_RNG_WALK=reference: every NoiseInjection draws its own map, in the reference's order
        b = latent.shape[0]
        run = [i for i in range(self.num_layers)
               if (i == 0 and start_layer == 0) or (i > 0 and max(start_layer, 1) <= (i + 1) // 2 and
                                                    ((i + 1) // 2 <= end_layer or (i + 1) // 2 == start_layer))]
        if not run:
            return noise
        sizes = [b * (2 ** ((i + 5) // 2)) ** 2 for i in run]
        flat = torch.empty(sum(sizes), device=latent.device, dtype=torch.float32).normal_()
        out, o = list(noise), 0
        for i, n in zip(run, sizes):
            r = 2 ** ((i + 5) // 2)
            out[i] = flat[o:o + n].view(b, 1, r, r)
            o += n
        return out

    def _batch_styles(self, latent):
        """Modulation (and demodulation) coefficients of every layer from W+ in two launches
        (hf_style_batch_f32) instead of one or two per layer; handed to the layers through
        ModulatedConv2d._coeffs for the duration of this forward.  Rows as in :529-545."""
        if latent.ndim != 3 or latent.shape[1] < self.n_latent or not latent.is_cuda:
            return []
        convs, rows = [self.conv1.conv, self.to_rgb1.conv], [0, 1]
        for block in range(1, self.log_size - 1):
            i = 2 * block - 1
            convs += [self.convs[2 * block - 2].conv, self.convs[2 * block - 1].conv, self.to_rgbs[block - 1].conv]
            rows += [i, i + 1, i + 2]
        b = latent.shape[0]
        for c in convs:
            c.prepared()
        key = (b, latent.device, tuple(c.modulation.weight.data_ptr() for c in convs), tuple(c._prep[1].data_ptr() for c in convs))
        cache = self.__dict__.setdefault("_style_jobs", {})  # one table per (batch, parameter storage)
        if key not in cache:
            if torch.cuda.is_current_stream_capturing():
                return []  # the table is uploaded from the host: never inside a graph capture (warm up first)
            if len(cache) > 16:
                cache.clear()
            table, layout, total = M.style_job_table(convs, rows, b, latent.device)
            cache[key] = (table, layout, total, max(c.in_channel for c in convs),
                          max(c.out_channel for c in convs if c.demodulate))
        table, layout, total, mci, mco = cache[key]
        lat = latent if (latent.dtype == torch.float32 and latent.stride(-1) == 1) else latent.float().contiguous()
        res = M.style_batch(lib(), stream(), lat, table, layout, total, mci, mco)
        for conv, row, (sv, dv) in zip(convs, rows, res):
            view = latent[:, row]
            conv._coeffs = ((view.data_ptr(), tuple(view.shape), tuple(view.stride())), (conv.prepared()[0], sv, dv))
        return convs

    def _run_layers(self, latent, noise, layer_in, skip, start_layer, end_layer, return_latents):
        out = self.input(latent)
        if start_layer == 0:
            out = self.conv1(out, latent[:, 0], noise=noise[0])
            skip = self.to_rgb1(out, latent[:, 1])
        if end_layer == 0:
            return out, skip
        i = 1
        split_in = up_coeffs = None  # SplitActivation of `out` for the next block's transposed conv (fast path)
        for block in range(1, self.log_size - 1):
            conv_up, conv_same, to_rgb = self.convs[2 * block - 2], self.convs[2 * block - 1], self.to_rgbs[block - 1]
            if block < start_layer:
                pass
            elif block != start_layer and block > end_layer:
                return out, skip
            else:
                src = layer_in if block == start_layer else out
                if split_in is not None and block != start_layer:
                    src = split_in  # the fp32 `out` may not even exist (see want_out below)
                rgb_style = latent[:, i + 2]
                cmid, csame = conv_up.conv.out_channel, conv_same.conv.out_channel
                h2, w2 = 2 * src.shape[2], 2 * src.shape[3]
                fast_mode = conv_precision() != "f32"
                nb_ = src.shape[0]
                if (fast_mode and cmid % 16 == 0 and M.modconv3x3_f16_supported(cmid, csame, h2, w2, batch=nb_)
                        and not _observed(conv_up, conv_same)):
                    # fast path: activations travel between the convs pre-modulated, split into fp16
                    # pairs and K-blocked (M.SplitActivation) - conv_up's blur pass writes conv_same's
                    # input, conv_same's epilogue writes the next block's conv_up input; an fp32
                    # activation is only materialised where something else reads it
                    coeffs = conv_same.conv.style_coefficients(latent[:, i + 1])
                    split = conv_up.forward_split(src, latent[:, i], noise[2 * block - 1], coeffs[1],
                                                  coeffs=up_coeffs if isinstance(src, M.SplitActivation) else None)
                    rgb = to_rgb.coefficients(rgb_style) if M.torgb_fusable(cmid, csame, h2, w2) else None
                    nb = block + 1
                    is_last = block == self.log_size - 2
                    s_up = None
                    if not is_last and nb <= end_layer:
                        nup, nsame = self.convs[2 * nb - 2], self.convs[2 * nb - 1]
                        if (csame % 16 == 0 and M.modconv3x3_up_f16_supported(csame, nup.conv.out_channel, h2, w2, batch=nb_)
                                and nup.conv.out_channel % 16 == 0
                                and M.modconv3x3_f16_supported(nup.conv.out_channel, nsame.conv.out_channel, 2 * h2, 2 * w2, batch=nb_)
                                and not _observed(nup, nsame)):
                            up_coeffs = nup.conv.style_coefficients(latent[:, i + 2])
                            s_up = up_coeffs[1]
                    fused_rgb = rgb is not None and not _observed(to_rgb)
                    # fp32 activation: read by a stand-alone ToRGB, returned on an early exit, or fed
                    # to the next block as a plain tensor
                    want_out = not fused_rgb or (not is_last and s_up is None)
                    if (fused_rgb and not want_out and s_up is None and skip is not None and M.image_fusable(cmid, csame, h2, w2)
                            and conv_same.activate.bias is not None):
                        # the last layer: its epilogue finishes ToRGB (bias + upsampled skip) - no raw product, no finishing launch
                        image = conv_same.forward_image(split, coeffs, noise[2 * block], rgb, to_rgb, skip)
                        if image is not None:
                            out, split_in, skip = None, None, image
                            i += 2
                            continue
                    out, raw, split_in = conv_same.forward_from_split(split, coeffs, noise[2 * block],
                                                                      rgb=rgb if fused_rgb else None, want_out=want_out,
                                                                      split_for=s_up)
                else:
                    assert not isinstance(src, M.SplitActivation)  # a split is only produced for a fast block
                    split_in = raw = None
                    out = conv_up(src, latent[:, i], noise=noise[2 * block - 1])
                    if conv_same.conv.fuses_torgb(out) and not _observed(conv_same, to_rgb):  # ToRGB's 1x1 conv in the epilogue
                        out, raw = conv_same.forward_rgb(out, latent[:, i + 1], noise[2 * block], to_rgb.coefficients(rgb_style))
                    else:
                        out = conv_same(out, latent[:, i + 1], noise=noise[2 * block])
                # the raw 1x1 product travels as an explicit value from the producing conv to its ToRGB
                skip = to_rgb.finish(raw, skip) if raw is not None else to_rgb(out, rgb_style, skip)
            i += 2
        image = skip
        return (image, latent) if return_latents else (image, None)
