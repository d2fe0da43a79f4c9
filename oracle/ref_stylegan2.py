"""CPU oracle: restatement of the reference StyleGAN2 generator forward.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Functional restatement, in
plain torch CPU fp32 ops, of the algorithm in the reference
(all file:line are relative to /root/reference):

  upfirdn2d            models/stylegan2/op/upfirdn2d.py:145-200 (native path),
                       CUDA spec models/stylegan2/op/upfirdn2d_kernel.cu:107-207
  fused_leaky_relu     models/stylegan2/op/fused_act.py:85-96,
                       CUDA spec fused_bias_act_kernel.cu:18-49 (act*10+grad == 30)
  equal_linear         models/stylegan2/model.py:134-168
  modulated_conv2d     models/stylegan2/model.py:238-279
  styled_conv          models/stylegan2/model.py:309-343 (+ NoiseInjection :282-293)
  to_rgb               models/stylegan2/model.py:346-365 (+ Upsample :35-53)
  generator_forward    models/stylegan2/model.py:477-565

Arithmetic lives in PyTorch ATen (third party; reference pins torch 1.13.1,
this container has 2.10 CPU).  Parity of this file against the imported
reference is checked by oracle/make_golden.py (bit-level report) and pinned by
the vectors under tests/golden/.

Parameters are passed as a flat mapping with the reference's state-dict keys
(e.g. ``convs.14.conv.weight``), so the same dict drives the reference module,
this oracle and the HIP path.
"""
import math

import torch
import torch.nn.functional as F

SQRT2 = 2 ** 0.5


# ----------------------------------------------------------------------------
# operator level
# ----------------------------------------------------------------------------

def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """x [N,C,H,W], kernel [kh,kw]; zero-insert x`up`, pad (negative = crop),
    true convolution with `kernel` (i.e. correlation with the flipped kernel),
    keep every `down`-th sample.  Same pad for both axes like the reference
    wrapper (op/upfirdn2d.py:145-156)."""
    n, c, h, w = x.shape
    kh, kw = kernel.shape
    p0, p1 = int(pad[0]), int(pad[1])
    z = x.reshape(n * c, 1, h, w)
    if up > 1:
        u = z.new_zeros(n * c, 1, h * up, w * up)
        u[:, :, ::up, ::up] = z
        z = u
    z = F.pad(z, [p0, p1, p0, p1])  # F.pad crops for negative values
    z = F.conv2d(z, torch.flip(kernel, [0, 1]).reshape(1, 1, kh, kw).to(z.dtype))
    z = z[:, :, ::down, ::down]
    return z.reshape(n, c, z.shape[2], z.shape[3])


def upfirdn2d_loops(x, kernel, up=1, down=1, pad=(0, 0)):
    """Definition-level triple loop (tiny inputs only): independent statement of
    upfirdn2d_kernel.cu:49-105 used to cross-check `upfirdn2d` above."""
    import numpy as np

    x = x.detach().double().numpy()
    k = kernel.detach().double().numpy()
    n, c, h, w = x.shape
    kh, kw = k.shape
    p0, p1 = pad
    out_h = (h * up + p0 + p1 - kh) // down + 1
    out_w = (w * up + p0 + p1 - kw) // down + 1
    out = np.zeros((n, c, out_h, out_w))
    for oy in range(out_h):
        for ox in range(out_w):
            acc = np.zeros((n, c))
            for ky in range(kh):
                for kx in range(kw):
                    # position in the zero-inserted, padded signal
                    uy = oy * down + ky - p0
                    ux = ox * down + kx - p0
                    if uy < 0 or ux < 0 or uy % up or ux % up:
                        continue
                    iy, ix = uy // up, ux // up
                    if iy >= h or ix >= w:
                        continue
                    acc += x[:, :, iy, ix] * k[kh - 1 - ky, kw - 1 - kx]
            out[:, :, oy, ox] = acc
    return torch.from_numpy(out)


def fused_leaky_relu(x, bias, negative_slope=0.2, scale=SQRT2):
    """act(x + b[c]) * scale with b broadcast on dim 1."""
    shape = [1, -1] + [1] * (x.ndim - 2)
    return F.leaky_relu(x + bias.reshape(shape), negative_slope) * scale


def equal_linear(x, weight, bias, lr_mul=1.0):
    """model.py:153-163 without activation: x @ (W*scale)^T + b*lr_mul."""
    scale = (1.0 / math.sqrt(weight.shape[1])) * lr_mul
    return F.linear(x, weight * scale, bias * lr_mul)


def blur_kernel_1d_to_2d(k=(1, 3, 3, 1), gain=1.0):
    k = torch.tensor(k, dtype=torch.float32)
    k2 = k[None, :] * k[:, None]
    return k2 / k2.sum() * gain


def modulated_conv2d(x, style, weight, mod_weight, mod_bias, demodulate=True,
                     upsample=False, blur_kernel=None):
    """x [B,Cin,H,W], style [B,style_dim], weight [1,Cout,Cin,k,k].

    Per-sample weights, batch folded into conv groups, exactly as
    model.py:238-279.  For upsample: conv_transpose2d(stride 2) followed by the
    4x4 blur with pad (1,1) (pad from model.py:204-210 for k=3)."""
    b, cin, h, w = x.shape
    _, cout, _, k, _ = weight.shape
    s = equal_linear(style, mod_weight, mod_bias).reshape(b, 1, cin, 1, 1)
    wgt = (1.0 / math.sqrt(cin * k * k)) * weight * s
    if demodulate:
        d = torch.rsqrt(wgt.pow(2).sum([2, 3, 4]) + 1e-8)
        wgt = wgt * d.reshape(b, cout, 1, 1, 1)
    if upsample:
        wt = wgt.transpose(1, 2).reshape(b * cin, cout, k, k)
        y = F.conv_transpose2d(x.reshape(1, b * cin, h, w), wt, padding=0, stride=2, groups=b)
        y = y.reshape(b, cout, y.shape[2], y.shape[3])
        p = (4 - 2) - (k - 1)
        pad = ((p + 1) // 2 + 2 - 1, p // 2 + 1)
        if blur_kernel is None:
            blur_kernel = blur_kernel_1d_to_2d(gain=4.0)
        return upfirdn2d(y, blur_kernel, pad=pad)
    y = F.conv2d(x.reshape(1, b * cin, h, w), wgt.reshape(b * cout, cin, k, k), padding=k // 2, groups=b)
    return y.reshape(b, cout, y.shape[2], y.shape[3])


def styled_conv(P, prefix, x, style, noise, upsample):
    """conv -> + noise_weight*noise -> fused bias leaky relu (model.py:337-343)."""
    y = modulated_conv2d(
        x, style, P[f"{prefix}.conv.weight"], P[f"{prefix}.conv.modulation.weight"],
        P[f"{prefix}.conv.modulation.bias"], demodulate=True, upsample=upsample,
        blur_kernel=P.get(f"{prefix}.conv.blur.kernel"))
    if noise is None:
        raise ValueError("oracle requires explicit noise (RNG parity is undefined across devices)")
    y = y + P[f"{prefix}.noise.weight"] * noise
    return fused_leaky_relu(y, P[f"{prefix}.activate.bias"])


def to_rgb(P, prefix, x, style, skip):
    """1x1 modulated conv without demod + bias + upsampled skip (model.py:356-365)."""
    y = modulated_conv2d(
        x, style, P[f"{prefix}.conv.weight"], P[f"{prefix}.conv.modulation.weight"],
        P[f"{prefix}.conv.modulation.bias"], demodulate=False)
    y = y + P[f"{prefix}.bias"]
    if skip is not None:
        kern = P.get(f"{prefix}.upsample.kernel")
        if kern is None:
            kern = blur_kernel_1d_to_2d(gain=4.0)
        y = y + upfirdn2d(skip, kern, up=2, down=1, pad=(2, 1))
    return y


# ----------------------------------------------------------------------------
# generator level
# ----------------------------------------------------------------------------

def generator_forward(P, latent, noise, layer_in=None, skip=None, start_layer=0,
                      end_layer=8, log_size=10):
    """Layer-range executor, model.py:532-565.  `latent` is W+ [B,n_latent,512];
    `noise` a list of 2*(log_size-2)+1 tensors [1|B,1,H,W] (never None).
    Returns (image, None) after the last block or (feature, skip) on early exit."""
    n_blocks = log_size - 2
    out = P["input.input"].repeat(latent.shape[0], 1, 1, 1)
    if start_layer == 0:
        out = styled_conv(P, "conv1", out, latent[:, 0], noise[0], upsample=False)
        skip = to_rgb(P, "to_rgb1", out, latent[:, 1], None)
    if end_layer == 0:
        return out, skip
    i = 1
    for blk in range(1, n_blocks + 1):
        c_up, c_same, rgb = f"convs.{2 * blk - 2}", f"convs.{2 * blk - 1}", f"to_rgbs.{blk - 1}"
        if blk < start_layer:
            pass
        elif blk > end_layer and blk != start_layer:
            return out, skip
        else:
            src = layer_in if blk == start_layer else out
            out = styled_conv(P, c_up, src, latent[:, i], noise[2 * blk - 1], upsample=True)
            out = styled_conv(P, c_same, out, latent[:, i + 1], noise[2 * blk], upsample=False)
            skip = to_rgb(P, rgb, out, latent[:, i + 2], skip)
        i += 2
    return skip, None


def mapping_network(P, z, n_mlp=8, lr_mlp=0.01):
    """z -> w: PixelNorm (model.py:16-21) then n_mlp x EqualLinear(lr_mul=lr_mlp, 'fused_lrelu')
    (model.py:384-393; EqualLinear.forward :153-163 with activation: F.linear(x, W*scale) then
    fused_leaky_relu(out, bias*lr_mul), scale = lr_mul/sqrt(in_dim))."""
    x = z * torch.rsqrt(torch.mean(z ** 2, dim=1, keepdim=True) + 1e-8)
    for i in range(1, n_mlp + 1):
        w, b = P[f"style.{i}.weight"], P[f"style.{i}.bias"]
        scale = (1 / math.sqrt(w.shape[1])) * lr_mlp
        x = fused_leaky_relu(F.linear(x, w * scale), b * lr_mlp)
    return x


def generator_param_shapes(size=1024, style_dim=512, n_mlp=8, channel_multiplier=2):
    """State-dict key -> shape for Generator(size, style_dim, n_mlp, cm)
    (layout from model.py:368-452); 171 entries for the 1024 config."""
    ch = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * channel_multiplier,
          128: 128 * channel_multiplier, 256: 64 * channel_multiplier,
          512: 32 * channel_multiplier, 1024: 16 * channel_multiplier}
    log_size = int(math.log2(size))
    S = {}
    for i in range(1, n_mlp + 1):
        S[f"style.{i}.weight"] = (style_dim, style_dim)
        S[f"style.{i}.bias"] = (style_dim,)
    S["input.input"] = (1, ch[4], 4, 4)

    def styled(prefix, cin, cout, up):
        S[f"{prefix}.conv.weight"] = (1, cout, cin, 3, 3)
        if up:
            S[f"{prefix}.conv.blur.kernel"] = (4, 4)
        S[f"{prefix}.conv.modulation.weight"] = (cin, style_dim)
        S[f"{prefix}.conv.modulation.bias"] = (cin,)
        S[f"{prefix}.noise.weight"] = (1,)
        S[f"{prefix}.activate.bias"] = (cout,)

    def rgb(prefix, cin, up):
        S[f"{prefix}.bias"] = (1, 3, 1, 1)
        if up:
            S[f"{prefix}.upsample.kernel"] = (4, 4)
        S[f"{prefix}.conv.weight"] = (1, 3, cin, 1, 1)
        S[f"{prefix}.conv.modulation.weight"] = (cin, style_dim)
        S[f"{prefix}.conv.modulation.bias"] = (cin,)

    styled("conv1", ch[4], ch[4], False)
    rgb("to_rgb1", ch[4], False)
    cin = ch[4]
    for i in range(3, log_size + 1):  # module registration order: convs, to_rgbs, noises
        cout = ch[2 ** i]
        styled(f"convs.{2 * (i - 3)}", cin, cout, True)
        styled(f"convs.{2 * (i - 3) + 1}", cout, cout, False)
        cin = cout
    for i in range(3, log_size + 1):
        rgb(f"to_rgbs.{i - 3}", ch[2 ** i], True)
    for li in range((log_size - 2) * 2 + 1):
        res = (li + 5) // 2
        S[f"noises.noise_{li}"] = (1, 1, 2 ** res, 2 ** res)
    return S
