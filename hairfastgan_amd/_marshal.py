"""Argument marshalling between torch tensors and the C ABI (include/hairfast_hip.h).

Pure plumbing: shape arithmetic, output allocation through torch's caching
allocator, ``data_ptr()`` extraction and error-code checking.  There is no
arithmetic here and no alternative code path: every function ends in exactly one
(or, for the upsampling conv, two) calls into the shared library it is given.
The public ops in ``hairfastgan_amd.op`` / ``hairfastgan_amd.stylegan2`` call
these with the HIP library and the current HIP stream after checking that all
tensors live on the GPU.
"""
import functools
import os

import torch
import torch.nn.functional as F

from ._lib import check
from ._runtime import plan_batch

SQRT2 = 2 ** 0.5

# bench.py sets this to a list to bracket every modulated-conv launch with HIP events
# recorded on the launch stream: entries are (kernel label, algorithmic flops, start, end).
PROFILE = None


KERNEL_NAMES = {  # hf_debug_last_path() code -> kernel instantiation (csrc/modconv.hip dispatch)
    211: "conv_mfma_pipe<2,2,2,4>", 212: "conv_mfma_pipe<2,2,1,4>", 213: "conv_mfma_pipe<1,2,1,4>",
    214: "conv_mfma_pipe<2,2,2,2>", 215: "conv_mfma_pipe<1,1,2,2>", 216: "conv_mfma_pipe<1,4,1,4>",
    231: "conv_mfma_dma<2,2,2,4>", 232: "conv_mfma_dma<2,2,1,4>", 233: "conv_mfma_dma<1,2,1,4>", 234: "conv_mfma_dma<2,2,2,2>",
    244: "conv_mfma_pipe<2,2,2,2,stride2>", 245: "conv_mfma_pipe<1,1,2,2,stride2>",
    221: "conv_mfma_pipe<1,2,2,2,up>", 222: "conv_mfma_pipe<1,2,1,4,up>", 223: "conv_mfma_pipe<1,1,2,2,up>",
    224: "conv_mfma_pipe<1,2,2,4,up>", 225: "conv_mfma_pipe<1,1,1,4,up>", 300: "conv_mfma<1,1,2,2> split-K",
    # csrc/convh.hip (fp16 matrix cores; <CT_TILES,PG,WAVES_CO,WAVES_PX>)
    581: "conv_mfma_h<1,2,2,4,up,pre>", 583: "conv_mfma_h<1,2,1,8,up,pre>",
    571: "conv_mfma_h<1,2,2,4,pre>", 572: "conv_mfma_h<2,2,1,8,pre>", 573: "conv_mfma_h<1,2,1,8,pre>", 575: "conv_mfma_h<1,2,1,8,tw128,pre>",
    579: "conv_rows_h<32->32,strip64,pre>",
    551: "conv_mfma_h<1,2,2,4>", 552: "conv_mfma_h<2,2,1,8>", 553: "conv_mfma_h<1,2,1,8>", 555: "conv_mfma_h<1,2,1,8,tw128>",
    561: "conv_mfma_h<1,2,2,4,up>", 563: "conv_mfma_h<1,2,1,8,up>",
    573: "conv_mfma_h<1,2,1,8,up,fuse>", 593: "conv_mfma_h<1,2,1,8,up,pre,fuse>",
    # csrc/convh_enc.hip (encoder convs on the fp16 matrix cores)
    601: "conv_enc_h<64x256>", 602: "conv_enc_h<64x128,stride2>", 603: "conv_enc_h<64x128>", 604: "conv_enc_h<64x512>",
    605: "conv_enc_h<64x128,stride2,x4 tiles>",
    606: "conv_enc_h<64x128,stride2,x2 tiles>",
}


def _launch_profiled(lib, flops, fn, label=None, nbytes=0.0, tag=""):
    """bench.py only: brackets one launch with HIP events on the launch stream.  Entries are
    (kernel label, algorithmic flops, start, end, algorithmic bytes); `label` None = a modulated /
    plain conv whose instantiation hf_debug_last_path reports."""
    if PROFILE is None:
        return fn()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    if label is None:
        code = lib.hf_debug_last_path()
        label = KERNEL_NAMES.get(code, f"conv_mfma (general, code {code})")
        if tag and label.endswith(">"):
            label = label[:-1] + tag + ">"
    PROFILE.append((label, flops, e0, e1, nbytes))
    return r


def _p(t):
    return None if t is None else t.data_ptr()


def _c(t):
    """fp32 + contiguous (the reference takes .contiguous() copies defensively too,
    fused_bias_act_kernel.cu:58-60, upfirdn2d_kernel.cu:219-220)."""
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise TypeError(f"hairfastgan_amd kernels are fp32; got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def upfirdn2d(lib, st, x, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    x, kernel = _c(x), _c(kernel)
    n, c, h, w = x.shape
    kh, kw = kernel.shape
    out_h = (h * up_y + pad_y0 + pad_y1 - kh) // down_y + 1
    out_w = (w * up_x + pad_x0 + pad_x1 - kw) // down_x + 1
    out = x.new_empty((n, c, out_h, out_w))
    if out.numel():
        check(lib, lib.hf_upfirdn2d_f32(_p(out), _p(x), _p(kernel), n * c, h, w, kh, kw, up_x, up_y, down_x,
                                        down_y, pad_x0, pad_x1, pad_y0, pad_y1, st), "hf_upfirdn2d_f32")
    return out


def fused_bias_act(lib, st, x, bias, alpha, scale):
    x, bias = _c(x), _c(bias)
    out = torch.empty_like(x)
    n = x.numel()
    if n == 0:
        return out
    step_b = 1
    for d in x.shape[2:]:
        step_b *= d
    n_bias = bias.numel() if bias is not None else 0
    if bias is not None and x.ndim >= 2 and n_bias != x.shape[1]:
        raise ValueError("bias must have one entry per channel (dim 1)")
    check(lib, lib.hf_fused_bias_act_f32(_p(out), _p(x), _p(bias), n, max(n_bias, 1), step_b, alpha, scale, st),
          "hf_fused_bias_act_f32")
    return out


def _noise_args(noise, batch, hw):
    if noise is None:
        return None, 0
    noise = _c(noise)
    if noise.numel() == hw:
        return noise, 0
    if noise.numel() == batch * hw:
        return noise, hw
    raise ValueError(f"noise must be [1,1,H,W] or [B,1,H,W]; got {tuple(noise.shape)}")


def noise_bias_act(lib, st, x, noise, noise_w, bias, alpha=0.2, scale=SQRT2):
    x, bias, noise_w = _c(x), _c(bias), _c(noise_w)
    b, c, h, w = x.shape
    noise, nbs = _noise_args(noise, b, h * w)
    out = torch.empty_like(x)
    check(lib, lib.hf_noise_bias_act_f32(_p(out), _p(x), _p(noise), _p(noise_w), _p(bias), b, c, h * w, nbs,
                                         alpha, scale, st), "hf_noise_bias_act_f32")
    return out


def prepare_weights(lib, st, weight):
    """weight [1,cout,cin,k,k] -> (wt [k*k,cin,cout], wsq [cout,cin])."""
    weight = _c(weight)
    _, cout, cin, k, _ = weight.shape
    wt = weight.new_empty((k * k, cin, cout))
    wsq = weight.new_empty((cout, cin))
    check(lib, lib.hf_modconv_prepare_f32(_p(wt), _p(wsq), _p(weight), cout, cin, k, st), "hf_modconv_prepare_f32")
    return wt, wsq


def modulation(lib, st, style, mod_w, mod_b):
    """style: [B, style_dim], possibly a strided row view of W+ (latent[:, i])."""
    if style.dtype != torch.float32:
        raise TypeError("style must be fp32")
    if style.stride(-1) != 1:
        style = style.contiguous()
    b, sd = style.shape
    mod_w, mod_b = _c(mod_w), _c(mod_b)
    cin = mod_w.shape[0]
    s = style.new_empty((b, cin))
    check(lib, lib.hf_modulation_f32(_p(s), _p(style), style.stride(0) if b > 1 else sd, _p(mod_w), _p(mod_b),
                                     b, cin, sd, st), "hf_modulation_f32")
    return s


def style_job_table(convs, rows, batch, device):
    """Device job table (hf_style_job records, 7 int64 = 56 bytes each) + output layout for hf_style_batch_f32.
    convs: ModulatedConv2d-like objects with .modulation.weight/.bias, .in_channel, .out_channel,
    .demodulate and .prepared(); rows: latent row of each.  Offsets depend on the batch size."""
    recs, layout, ofs = [], [], 0
    for conv, row in zip(convs, rows):
        _, wsq = conv.prepared()
        cin, cout = conv.in_channel, conv.out_channel
        s_ofs = ofs
        ofs += batch * cin
        d_ofs = -1
        if conv.demodulate:
            d_ofs = ofs
            ofs += batch * cout
        mw, mb = conv.modulation.weight.detach(), conv.modulation.bias.detach()
        recs.append([mw.data_ptr(), mb.data_ptr(), wsq.data_ptr() if conv.demodulate else 0, cin | (cout << 32), row,
                     s_ofs, max(d_ofs, 0)])
        layout.append((s_ofs, cin, d_ofs, cout))
    table = torch.tensor(recs, dtype=torch.int64).to(device)
    return table, layout, ofs


def style_batch(lib, st, latent, table, layout, total, max_cin, max_cout):
    """All (s, d) pairs of a forward in two launches; returns [(s, d|None)] views per job."""
    if latent.dtype != torch.float32 or latent.stride(-1) != 1:
        latent = latent.float().contiguous()
    b, _, sd = latent.shape
    out = latent.new_empty(total)
    check(lib, lib.hf_style_batch_f32(_p(out), _p(latent), latent.stride(0), latent.stride(1), _p(table), len(layout), b, sd,
                                      max_cin, max_cout, st), "hf_style_batch_f32")
    res = []
    for s_ofs, cin, d_ofs, cout in layout:
        s = out[s_ofs:s_ofs + b * cin].view(b, cin)
        d = out[d_ofs:d_ofs + b * cout].view(b, cout) if d_ofs >= 0 else None
        res.append((s, d))
    return res


def demod(lib, st, s, wsq):
    b, cin = s.shape
    cout = wsq.shape[0]
    d = s.new_empty((b, cout))
    check(lib, lib.hf_demod_f32(_p(d), _p(s), _p(wsq), b, cin, cout, st), "hf_demod_f32")
    return d


def style_normalize(lib, st, s, d):
    """In-place power-of-two range normalisation of a layer's (s, d) pair (hf_style_normalize_f32)."""
    b, cin = s.shape
    check(lib, lib.hf_style_normalize_f32(_p(s), _p(d), b, cin, d.shape[1], st), "hf_style_normalize_f32")


def f16_overflow_count(lib, reset=False):
    """Elements the fp16 (hi, lo) split clamped since the last reset (synchronises the device)."""
    return int(lib.hf_f16_overflow_count(1 if reset else 0))


def _workspace(lib, like, b, cin, cout, h, w, up):
    """Split-K scratch for the small-plane layers (size dictated by the library)."""
    n = lib.hf_modconv_workspace_floats(b, cin, cout, h, w, 1 if up else 0)
    if n <= 0:
        return None, 0
    return like.new_empty((n,)), n


def modconv3x3(lib, st, x, wt, s, d, noise, noise_w, bias, alpha=0.2, scale=SQRT2):
    x = _c(x)
    b, cin, h, w = x.shape
    cout = wt.shape[2]
    noise, nbs = _noise_args(noise, b, h * w)
    out = x.new_empty((b, cout, h, w))
    noise_w, bias = _c(noise_w), _c(bias)
    ws, ws_n = _workspace(lib, x, b, cin, cout, h, w, False)
    code = _launch_profiled(
        lib, 2.0 * cin * cout * 9 * h * w * b,
        lambda: lib.hf_modconv3x3_f32(_p(out), _p(x), _p(wt), _p(s), _p(d), _p(noise), _p(noise_w), nbs, _p(bias),
                                      b, cin, cout, h, w, alpha, scale, _p(ws), ws_n, st))
    check(lib, code, "hf_modconv3x3_f32")
    return out


def split_weights_f16(lib, st, wt):
    """Prepared fp32 weights [9, cin, cout] -> (hi, lo) fp16 tensors in the K-contiguous
    layout of csrc/convh.hip ([cin/16][tap][2][cout][8])."""
    wt = _c(wt)
    taps, cin, cout = wt.shape
    if taps != 9 or cin % 16:
        raise ValueError(f"f16 MFMA path needs 3x3 weights with cin % 16 == 0; got {tuple(wt.shape)}")
    n = 9 * cin * cout
    # wt_hi carries a 16-byte trailer behind its n halves (2^-k of the power-of-two pre-scale)
    hi = torch.empty(n + 8, dtype=torch.float16, device=wt.device)[:n].view(cin // 16, 9, 2, cout, 8)
    lo = torch.empty((cin // 16, 9, 2, cout, 8), dtype=torch.float16, device=wt.device)
    check(lib, lib.hf_conv_split_weights_f16(_p(hi), _p(lo), _p(wt), cin, cout, st), "hf_conv_split_weights_f16")
    return hi, lo


def split_weights_unscale(wt_hi):
    """The 2^-k of the power-of-two pre-scale, read from the trailer behind wt_hi's halves (tests)."""
    n = wt_hi.numel()
    flat = torch.empty(0, dtype=torch.float16, device=wt_hi.device).set_(wt_hi.untyped_storage(), wt_hi.storage_offset(), (n + 8,))
    return float(flat[n:n + 2].view(torch.float32)[0])


def modconv3x3_f16_supported(cin, cout, h, w, batch=None):
    """Shapes hf_modconv3x3_f16_f32 takes (include/hairfast_hip.h).  batch given: also whether it PAYS - a launch of
    fewer than 2048 pixels in total (a batch-1 32^2 layer: 4 tiles x 8 channel tiles) leaves the chip to 32 blocks that
    each walk the whole K loop (94 us); the fp32 kernels split K over the CUs instead (70 us; tools/probes/tower.py)."""
    if batch is not None and plan_batch(batch) * h * w < 2048:
        return False
    return cin % 16 == 0 and w >= 32 and (h >= 8 if cout % 64 == 0 else (cout % 32 == 0 and h >= 16))


def torgb_fusable(cin, cout, h, w):
    """Layers whose ToRGB hf_modconv3x3_f16_rgb_f32 computes in the conv epilogue."""
    if not (cout % 32 == 0 and modconv3x3_f16_supported(cin, cout, h, w)):
        return False
    # one slab (cout 32 / 64) from 512 pixels; several slabs need the 64-channel-per-wave tile shape, which only
    # fills the chip from 64^2 planes (a 32^2 layer on it measured 2x slower than on its own tile shape)
    return h * w >= (512 if cout in (32, 64) else 4096)


def image_fusable(cin, cout, h, w):
    """The generator's last StyledConv whose epilogue finishes ToRGB (hf_modconv3x3_f16_pre_image_f32: the row pipeline's
    shapes).  The library may still decline (debug dispatch / tuning switches): modconv3x3_f16_pre_image then returns None."""
    return cin == 32 and cout == 32 and w % 64 == 0 and h % 8 == 0 and os.environ.get("HAIRFAST_IMAGE_FUSE", "1") != "0"


def modconv3x3_f16_pre_image(lib, st, act, wt_hi, wt_lo, nterms, d, noise, noise_w, bias, rgb, rgb_bias, skip, up_kernel,
                             alpha=0.2, scale=SQRT2):
    """hf_modconv3x3_f16_pre_image_f32: same-resolution 3x3 conv on a SplitActivation + its complete ToRGB (1x1 modulated conv
    + bias + upsampled skip) in one launch - the image [B,3,H,W], or None when the library does not take the shape (the caller
    runs modconv3x3_f16_pre(rgb=...) + torgb: the same bits)."""
    b, cin, h, w = act.shape
    cout = wt_hi.shape[3]
    skip = _c(skip)
    if tuple(skip.shape) != (b, 3, h // 2, w // 2):
        raise ValueError(f"skip must be [B,3,H/2,W/2]; got {tuple(skip.shape)} for a {h}x{w} layer")
    noise, nbs = _noise_args(noise, b, h * w)
    image = torch.empty((b, 3, h, w), dtype=torch.float32, device=act.hi.device)
    rgb_wt, rgb_s = _c(rgb[0]), _c(rgb[1])
    nparts = 2 if nterms == 3 else 1
    nb = float(b) * h * w * (2.0 * nparts * cin + 12.0 + 3.0) + 2.0 * nparts * 9 * cin * cout  # split input, image, skip, weights: once
    code = _launch_profiled(
        lib, 2.0 * cin * cout * 9 * h * w * b,
        lambda: lib.hf_modconv3x3_f16_pre_image_f32(_p(image), _p(act.hi), _p(act.lo), _p(wt_hi), _p(wt_lo), nterms, _p(d), _p(noise),
                                                    _p(_c(noise_w)), nbs, _p(_c(bias)), b, cin, cout, h, w, alpha, scale, _p(rgb_wt),
                                                    _p(rgb_s), _p(_c(rgb_bias)), _p(skip), _p(_c(up_kernel)), st), nbytes=nb)
    if code == -1:  # HF_E_INVALID: not a row-pipeline launch
        return None
    check(lib, code, "hf_modconv3x3_f16_pre_image_f32")
    return image


def torgb_slabs(cout):
    """Slabs of the fused ToRGB's raw tensor [B, 3*slabs, H, W] (hf_modconv3x3_f16_rgb_slabs)."""
    return cout // 64 if cout % 64 == 0 else cout // 32


def modconv3x3_f16(lib, st, x, wt_hi, wt_lo, nterms, s, d, noise, noise_w, bias, alpha=0.2, scale=SQRT2, rgb=None):
    """hf_modconv3x3_f32 on the fp16 matrix cores (nterms 3: split operands, fp32-class
    accuracy; nterms 1: fp16 operands); fp32 tensors and accumulation.
    rgb = (rgb_wt [1,cout,3], rgb_s [B,cout]): also returns ToRGB's raw 1x1 modulated conv of the
    output, computed in the epilogue (hf_modconv3x3_f16_rgb_f32)."""
    x = _c(x)
    b, cin, h, w = x.shape
    cout = wt_hi.shape[3]
    noise, nbs = _noise_args(noise, b, h * w)
    out = x.new_empty((b, cout, h, w))
    noise_w, bias = _c(noise_w), _c(bias)
    if rgb is not None:
        rgb_wt, rgb_s = _c(rgb[0]), _c(rgb[1])
        raw = x.new_empty((b, 3 * torgb_slabs(cout), h, w))
        code = _launch_profiled(
            lib, 2.0 * cin * cout * 9 * h * w * b,
            lambda: lib.hf_modconv3x3_f16_rgb_f32(_p(out), _p(x), _p(wt_hi), _p(wt_lo), nterms, _p(s), _p(d), _p(noise),
                                                  _p(noise_w), nbs, _p(bias), b, cin, cout, h, w, alpha, scale,
                                                  _p(raw), _p(rgb_wt), _p(rgb_s), st))
        check(lib, code, "hf_modconv3x3_f16_rgb_f32")
        return out, raw
    code = _launch_profiled(
        lib, 2.0 * cin * cout * 9 * h * w * b,
        lambda: lib.hf_modconv3x3_f16_f32(_p(out), _p(x), _p(wt_hi), _p(wt_lo), nterms, _p(s), _p(d), _p(noise),
                                          _p(noise_w), nbs, _p(bias), b, cin, cout, h, w, alpha, scale, st))
    check(lib, code, "hf_modconv3x3_f16_f32")
    return out


def split_activation_reference(x, s):
    """torch statement of the producer-side split (tests / tools): s*x -> (hi, lo) K-blocked."""
    b, c, h, w = x.shape
    v = x if s is None else x * s[:, :, None, None]
    hi = v.half()
    lo = (v - hi.float()).half()
    blk = lambda t: t.reshape(b, c // 8, 8, h, w).permute(0, 1, 3, 4, 2).contiguous()  # noqa: E731
    return blk(hi), blk(lo)


def modconv3x3_f16_pre(lib, st, act, wt_hi, wt_lo, nterms, d, noise, noise_w, bias, alpha=0.2, scale=SQRT2, rgb=None,
                       want_out=True, split_for=None):
    """hf_modconv3x3_f16_pre_f32: same-resolution 3x3 conv on the fp16 matrix cores whose input is a
    SplitActivation (modulation already applied by the producer).
    rgb = (rgb_wt, rgb_s): fused ToRGB raw product (see modconv3x3_f16);
    split_for = s_next [B,cout]: the epilogue also writes a SplitActivation of s_next*out for the next
    layer's transposed conv; want_out=False: the fp32 activation itself is not written.
    Returns out, or (out, raw), (out, split), (out, raw, split) in that order of optional parts."""
    b, cin, h, w = act.shape
    cout = wt_hi.shape[3]
    dev = act.hi.device
    noise, nbs = _noise_args(noise, b, h * w)
    if not want_out and rgb is None and split_for is None:
        raise ValueError("want_out=False needs another consumer (rgb= or split_for=)")
    out = torch.empty((b, cout, h, w), dtype=torch.float32, device=dev) if want_out else None
    noise_w, bias = _c(noise_w), _c(bias)
    raw = rgb_wt = rgb_s = sh = sl = s_next = None
    if rgb is not None:
        rgb_wt, rgb_s = _c(rgb[0]), _c(rgb[1])
        raw = torch.empty((b, 3 * torgb_slabs(cout), h, w), dtype=torch.float32, device=dev)
    if split_for is not None:
        s_next = _c(split_for)
        sh = torch.empty((b, cout // 8, h, w, 8), dtype=torch.float16, device=dev)
        sl = torch.empty_like(sh) if nterms == 3 else None  # plain fp16 consumer: no lo part
    # algorithmic HBM bytes (bench.py: traffic / algorithmic): the split input once, every output form once, the weights once
    nparts = 2 if nterms == 3 else 1
    nb = (float(b) * h * w * (2.0 * nparts * cin + (4.0 * cout if want_out else 0.0) + (2.0 * nparts * cout if split_for is not None else 0.0)
                              + (12.0 * torgb_slabs(cout) if rgb is not None else 0.0)) + 2.0 * nparts * 9 * cin * cout)
    code = _launch_profiled(
        lib, 2.0 * cin * cout * 9 * h * w * b,
        lambda: lib.hf_modconv3x3_f16_pre_f32(_p(out), _p(act.hi), _p(act.lo), _p(wt_hi), _p(wt_lo), nterms, _p(d), _p(noise),
                                              _p(noise_w), nbs, _p(bias), b, cin, cout, h, w, alpha, scale, _p(raw),
                                              _p(rgb_wt), _p(rgb_s), _p(sh), _p(sl), _p(s_next), st), nbytes=nb)
    check(lib, code, "hf_modconv3x3_f16_pre_f32")
    res = [out]
    if rgb is not None:
        res.append(raw)
    if split_for is not None:
        res.append(SplitActivation(sh, sl, None))
    return res[0] if len(res) == 1 else tuple(res)


def split_weights_small(lib, st, wt):
    """Prepared weights wt [9, cin, cout] -> the (hi, lo) blobs hf_modconv3x3_small_f16_f32 takes: the nine taps as the
    rows of ONE 1x1 GEMM, [1][cin][tap*cout + co], split by hf_conv_split_weights_f16_taps(taps = 1)."""
    taps, cin, cout = wt.shape
    w1 = _c(wt).permute(1, 0, 2).reshape(1, cin, taps * cout).contiguous()
    return conv_split_weights_f16(lib, st, w1)


def modconv3x3_small_supported(cin, cout, h, w, batch, upsample=False):
    """Shapes hf_modconv3x3_small_f16_f32 takes AND pays for (tools/probes/tower.py, 512 -> 512 channels, batch 1 / 3 / 8):
    same resolution: 256 < batch*h*w <= 2048 pixels per launch (below, the three launches of the tap-GEMM form cost more than
    the fp32 split-K kernel's two; above, the tiled fp16-core conv fills the chip) - 8^2 at batch 8: 43 -> 30 us, 16^2: 98 ->
    81 us, 32^2 at batch 1: 70 -> 56 us; transposed: inputs up to 8^2 at any batch (4 -> 8: 72 -> 56 us, 8 -> 16: 141 -> 58 us
    at batch 8; 62 -> 38 / 68 -> 43 us at batch 1) and 16^2 inputs up to 1024 pixels per launch (batch 3: 110 -> 88 us)."""
    if cin % 32 or cout % 64 or h * w > 1024:
        return False
    n = plan_batch(batch) * h * w  # batch-invariant mode: the dispatch of ONE sample (_runtime.plan_batch)
    if upsample:
        return h * w <= 64 or (h * w <= 256 and n <= 1024)
    # n = 256 from tiny planes (4^2 at batch 16, 8^2 at batch 4): the fp32 split-K kernel takes 180 us there, the tap GEMM 60
    # 8^2 planes from n = 192 = the canonical batch 3 of the batch-invariant plans (round 6: at batch 8 the fp32 split-K family that
    # batch 3 alone would take costs 53 us, the tap GEMM 29; at batch 3 itself 29 vs 31 - tools/probes/tower.py, profiles/r06an_*)
    return 256 < n <= 2048 or (n == 256 and h * w <= 64) or (h * w == 64 and n >= 192 and n <= 2048)


def conv3x3_small_supported(cin, cout, h, w, batch):
    """Plain (unmodulated) 3x3 convs worth the tap-GEMM form: planes the tiled fp16 kernel does not take, weights of 1 MB
    and more (below, the fp32 split-K kernel's two launches win), at most 4096 pixels per launch."""
    return cin % 32 == 0 and cout % 64 == 0 and h * w <= 256 and plan_batch(batch) * h * w <= 4096 and cin * cout >= 512 * 512


def modconv3x3_small(lib, st, x, w9, nterms, s, d, noise, noise_w, bias, cout, alpha=0.2, scale=SQRT2, upsample=False):
    """hf_modconv3x3_small_f16_f32.  upsample: returns the (2h+1) x pitch intermediate [B,cout,2h+1,pitch] (demodulated)."""
    x = _c(x)
    b, cin, h, w = x.shape
    hi, lo = w9
    n = lib.hf_modconv3x3_small_workspace_floats(b, cin, cout, h, w)
    ws = x.new_empty((max(n, 1),))
    if upsample:
        pitch = lib.hf_modconv_up_pitch(w)
        out = x.new_empty((b, cout, 2 * h + 1, pitch))
        noise = bias = noise_w = None
        nbs = 0
    else:
        pitch = 0
        out = x.new_empty((b, cout, h, w))
        noise, nbs = _noise_args(noise, b, h * w)
    code = _launch_profiled(
        lib, 2.0 * cin * cout * 9 * h * w * b,
        lambda: lib.hf_modconv3x3_small_f16_f32(_p(out), _p(x), _p(hi), _p(lo), nterms, _p(_c(s)), _p(_c(d)), _p(noise),
                                                _p(_c(noise_w)), nbs, _p(_c(bias)), b, cin, cout, h, w, alpha, scale,
                                                1 if upsample else 0, pitch, _p(ws), n, st),
        label="gemm_h tap-GEMM (small planes)")
    check(lib, code, "hf_modconv3x3_small_f16_f32")
    return out


SMALL_UP_FUSED = os.environ.get("HAIRFAST_SMALL_UP_FUSED", "1") != "0"  # 0: tap GEMM, combine pass, blur pass (A/B, tests)


def small_up_blur_supported(h, w):
    """hf_modconv3x3_small_up_blur_f16_f32's LDS bound: 8 channels x ((2h+3) x (2w+3) + 9 h w) floats."""
    return (8 * (2 * h + 3) * (2 * w + 3) + 72 * h * w) * 4 <= 150 * 1024


def modconv3x3_small_up_blur(lib, st, x, w9, nterms, s, d, blur_kernel, noise, noise_w, bias, cout, alpha=0.2, scale=SQRT2,
                             split_for=None):
    """hf_modconv3x3_small_up_blur_f16_f32: the small-plane upsampling StyledConv as tap GEMM + one combine / blur / tail kernel.
    split_for as in modconv3x3_up: None -> the fp32 activation [B,cout,2h,2w]; (key, s_next[, want_lo]) -> SplitActivation."""
    x = _c(x)
    b, cin, h, w = x.shape
    hi, lo = w9
    n = lib.hf_modconv3x3_small_workspace_floats(b, cin, cout, h, w)
    ws = x.new_empty((max(n, 1),))
    noise, nbs = _noise_args(noise, b, 4 * h * w)
    out = sh = sl = s_next = key = None
    if split_for is not None:
        key, s_next = split_for[0], _c(split_for[1])
        want_lo = split_for[2] if len(split_for) > 2 else True
        sh = torch.empty((b, cout // 8, 2 * h, 2 * w, 8), dtype=torch.float16, device=x.device)
        sl = torch.empty_like(sh) if want_lo else None
    else:
        out = x.new_empty((b, cout, 2 * h, 2 * w))
    code = _launch_profiled(
        lib, 2.0 * cin * cout * 9 * h * w * b,
        lambda: lib.hf_modconv3x3_small_up_blur_f16_f32(_p(out), _p(sh), _p(sl), _p(x), _p(hi), _p(lo), nterms, _p(_c(s)), _p(_c(d)),
                                                        _p(_c(blur_kernel)), _p(noise), _p(_c(noise_w)), nbs, _p(_c(bias)),
                                                        _p(s_next), b, cin, cout, h, w, alpha, scale, _p(ws), n, st),
        label="gemm_h tap-GEMM (small planes)")
    check(lib, code, "hf_modconv3x3_small_up_blur_f16_f32")
    return out if split_for is None else SplitActivation(sh, sl, key)


def modconv3x3_up_f16_supported(cin, cout, h, w, batch=None):
    """Shapes hf_modconv3x3_up_f16_f32 takes (include/hairfast_hip.h); batch: see modconv3x3_f16_supported (a batch-1
    16^2 -> 32^2 layer: 84 us on the fp32 split-K kernels, 106 us here)."""
    if batch is not None and plan_batch(batch) * h * w < 512:
        return False
    return cin % 16 == 0 and cout % 32 == 0 and h * w >= (256 if cout % 64 == 0 else 512) and min(h, w) >= 2


class SplitActivation:
    """An activation handed from a producer to a 3x3 conv on the fp16 matrix cores without an fp32
    round trip: s_next * y split into fp16 (hi, lo) and K-blocked [B, C/8, H, W, 8] (csrc/convh.hip)."""

    def __init__(self, hi, lo, key):
        self.hi, self.lo, self.key = hi, lo, key
        b, cb, h, w, _ = hi.shape
        self.shape = (b, cb * 8, h, w)


def modconv3x3_up(lib, st, x, wt, s, d, blur_kernel, noise, noise_w, bias, alpha=0.2, scale=SQRT2, f16=None,
                  split_for=None, small=None):
    """conv_transpose(stride 2) -> [B,cout,2h+1,2w+1] scratch -> blur+noise+bias+act -> [B,cout,2h,2w].
    f16 = (wt_hi, wt_lo, nterms): part 1 on the fp16 matrix cores (hf_modconv3x3_up_f16_f32).
    split_for = (key, s_next [B,cout]): part 2 writes a SplitActivation for the conv whose modulation
    is s_next instead of the fp32 tensor (hf_blur_noise_bias_act_split_f16).
    small = (w9, nterms): part 1 as the small-plane tap-GEMM (hf_modconv3x3_small_f16_f32, upsample form)."""
    pre = isinstance(x, SplitActivation)
    if not pre:
        x = _c(x)
    b, cin, h, w = x.shape
    cout = wt.shape[2]
    pitch = lib.hf_modconv_up_pitch(w)  # rows padded to a multiple of 4 floats (aligned 16 B loads in the blur)
    if small is not None and not pre and SMALL_UP_FUSED and small_up_blur_supported(h, w):
        # round 6: tap GEMM + ONE combine / blur / tail kernel (the (2h+1)^2 intermediate stays in LDS) - same bits
        return modconv3x3_small_up_blur(lib, st, x, small[0], small[1], s, d, blur_kernel, noise, noise_w, bias, cout, alpha, scale,
                                        split_for=split_for)
    if small is not None and not pre:
        tmp = modconv3x3_small(lib, st, x, small[0], small[1], s, d, None, None, None, cout, upsample=True)
    else:
        tmp = torch.empty((b, cout, 2 * h + 1, pitch), dtype=torch.float32, device=(x.hi if pre else x).device)
    if small is not None and not pre:
        pass
    elif pre:  # pre-split input (modulation already applied by the producer): hf_modconv3x3_up_f16_pre_f32
        hi, lo, nterms = f16
        code = _launch_profiled(
            lib, 2.0 * cin * cout * 9 * h * w * b,
            lambda: lib.hf_modconv3x3_up_f16_pre_f32(_p(tmp), _p(x.hi), _p(x.lo), _p(hi), _p(lo), nterms, _p(d), b, cin, cout,
                                                     h, w, pitch, st))
        check(lib, code, "hf_modconv3x3_up_f16_pre_f32")
    elif f16 is not None:
        hi, lo, nterms = f16
        code = _launch_profiled(
            lib, 2.0 * cin * cout * 9 * h * w * b,
            lambda: lib.hf_modconv3x3_up_f16_f32(_p(tmp), _p(x), _p(hi), _p(lo), nterms, _p(s), _p(d), b, cin, cout, h, w,
                                                 pitch, st))
        check(lib, code, "hf_modconv3x3_up_f16_f32")
    else:
        ws, ws_n = _workspace(lib, x, b, cin, cout, h, w, True)
        code = _launch_profiled(
            lib, 2.0 * cin * cout * 9 * h * w * b,
            lambda: lib.hf_modconv3x3_up_f32(_p(tmp), _p(x), _p(wt), _p(s), _p(d), b, cin, cout, h, w, pitch, _p(ws),
                                             ws_n, st))
        check(lib, code, "hf_modconv3x3_up_f32")
    noise, nbs = _noise_args(noise, b, 4 * h * w)
    if split_for is not None:
        key, s_next = split_for[0], split_for[1]
        want_lo = split_for[2] if len(split_for) > 2 else True  # False: consumer with plain fp16 operands
        hi = torch.empty((b, cout // 8, 2 * h, 2 * w, 8), dtype=torch.float16, device=tmp.device)
        lo = torch.empty_like(hi) if want_lo else None
        # algorithmic bytes: the (2h+1)(2w+1) fp32 intermediate read once, one 2-byte value per part written
        nb = float(b * cout) * (4.0 * (2 * h + 1) * (2 * w + 1) + (4.0 if want_lo else 2.0) * 4 * h * w)
        code = _launch_profiled(
            lib, 0.0,
            lambda: lib.hf_blur_noise_bias_act_split_f16(_p(hi), _p(lo), _p(tmp), _p(_c(blur_kernel)), _p(noise),
                                                         _p(_c(noise_w)), nbs, _p(_c(bias)), _p(_c(s_next)), b, cout,
                                                         2 * h + 1, 2 * w + 1, pitch, alpha, scale, st),
            label="blur4x4_split8", nbytes=nb)
        check(lib, code, "hf_blur_noise_bias_act_split_f16")
        return SplitActivation(hi, lo, key)
    out = tmp.new_empty((b, cout, 2 * h, 2 * w))
    nb = float(b * cout) * 4.0 * ((2 * h + 1) * (2 * w + 1) + 4 * h * w)
    code = _launch_profiled(
        lib, 0.0,
        lambda: lib.hf_blur_noise_bias_act_f32(_p(out), _p(tmp), _p(_c(blur_kernel)), _p(noise), _p(_c(noise_w)),
                                               nbs, _p(_c(bias)), b, cout, 2 * h + 1, 2 * w + 1, pitch, alpha, scale, st),
        label="blur4x4_noise_bias_act", nbytes=nb)
    check(lib, code, "hf_blur_noise_bias_act_f32")
    return out


def blur_factors(kernel4x4):
    """(kx, ky): the 1-D factors of a rank-1 4x4 blur kernel (k[r][j] = ky[r]*kx[j]) as ctypes float[4] arrays,
    or None when the kernel is not separable.  Reads the kernel on the host (synchronises): callers cache it."""
    import ctypes

    k = kernel4x4.detach().double().cpu()
    if tuple(k.shape) != (4, 4) or float(k[0, 0]) == 0.0:
        return None
    kx, ky = k[0, :].clone(), k[:, 0] / k[0, 0]
    if not torch.allclose(torch.outer(ky, kx), k, rtol=1e-6, atol=1e-12):
        return None
    arr = ctypes.c_float * 4
    return arr(*[float(v) for v in kx]), arr(*[float(v) for v in ky])


def modconv3x3_up_fused_supported(cin, cout, h, w):
    """Shapes hf_modconv3x3_up_blur_f16_f32 takes."""
    return cin % 16 == 0 and cout % 32 == 0 and h >= 16 and w >= 32


def modconv3x3_up_fused(lib, st, x, wt_hi, wt_lo, s, d, factors, noise, noise_w, bias, alpha=0.2, scale=SQRT2, split_for=None,
                        nterms=3):
    """hf_modconv3x3_up_blur_f16_f32: transposed conv + blur + noise + bias + lrelu in one kernel (nterms 3: f16x3; 1: plain
    fp16 operands - wt_lo is then not passed, a split output has no lo part).
    x: fp32 tensor (with s) or SplitActivation.  Returns the fp32 activation, or - split_for = s_next [B,cout] -
    the SplitActivation of s_next * activation for a pre-split consumer (one output form per launch)."""
    pre = isinstance(x, SplitActivation)
    if not pre:
        x = _c(x)
    b, cin, h, w = x.shape
    cout = wt_hi.shape[3]
    dev = x.hi.device if pre else x.device
    noise, nbs = _noise_args(noise, b, 4 * h * w)
    out = sh = sl = s_next = None
    if split_for is not None:
        s_next = _c(split_for)
        sh = torch.empty((b, cout // 8, 2 * h, 2 * w, 8), dtype=torch.float16, device=dev)
        sl = torch.empty_like(sh) if nterms == 3 else None
    else:
        out = torch.empty((b, cout, 2 * h, 2 * w), dtype=torch.float32, device=dev)
    kx, ky = factors
    nb_alg = float(b) * (4.0 * cin * h * w + 4.0 * cout * 4 * h * w + 4.0 * 4 * h * w) + 4.0 * 9 * cin * cout  # input, output, noise, weights
    code = _launch_profiled(
        lib, 2.0 * cin * cout * 9 * h * w * b,
        lambda: lib.hf_modconv3x3_up_blur_f16_f32(_p(out), _p(sh), _p(sl), None if pre else _p(x), _p(x.hi) if pre else None,
                                                  _p(x.lo) if (pre and nterms == 3) else None, _p(wt_hi),
                                                  _p(wt_lo) if nterms == 3 else None, None if pre else _p(s), _p(d),
                                                  kx, ky, _p(noise), _p(_c(noise_w)), nbs, _p(_c(bias)), _p(s_next), b, cin, cout,
                                                  h, w, alpha, scale, st), nbytes=nb_alg)
    check(lib, code, "hf_modconv3x3_up_blur_f16_f32")
    return out if split_for is None else SplitActivation(sh, sl, None)


def torgb(lib, st, x, wt, s, bias, skip, up_kernel):
    x = _c(x)
    b, cin, h, w = x.shape
    skip = _c(skip)
    if skip is not None and tuple(skip.shape) != (b, 3, h // 2, w // 2):
        raise ValueError(f"skip must be [B,3,H/2,W/2]; got {tuple(skip.shape)} for x {tuple(x.shape)}")
    out = x.new_empty((b, 3, h, w))
    nb = float(b) * 4.0 * h * w * (cin + 3 + (0.75 if skip is not None else 0.0))  # x read, rgb written, skip read
    code = _launch_profiled(
        lib, 0.0,
        lambda: lib.hf_torgb_f32(_p(out), _p(x), _p(wt), _p(s), _p(_c(bias)), _p(skip), _p(_c(up_kernel)), b, cin, h, w, st),
        label="torgb_kernel" if h * w > 4096 else "torgb_small_kernel", nbytes=nb)
    check(lib, code, "hf_torgb_f32")
    return out


# ----------------------------------------------------------------------------------------
# encoders
# ----------------------------------------------------------------------------------------
ACT_NONE, ACT_LRELU, ACT_PRELU, ACT_QGELU = 0, 1, 2, 3
ACT_RESIDUAL_FIRST = 16  # OR-ed into act: residual added before the activation (HF_ACT_RESIDUAL_FIRST)


def conv_prepare(lib, st, weight, scale=1.0):
    """Conv2d weight [cout,cin,k,k] -> wt [k*k,cin,cout]."""
    weight = _c(weight)
    cout, cin, k, _ = weight.shape
    wt = weight.new_empty((k * k, cin, cout))
    check(lib, lib.hf_conv_prepare_f32(_p(wt), _p(weight), cout, cin, k, scale, st), "hf_conv_prepare_f32")
    return wt


def bn_fold(lib, st, gamma, beta, mean, var, eps, conv_bias=None):
    gamma, beta, mean, var, conv_bias = _c(gamma), _c(beta), _c(mean), _c(var), _c(conv_bias)
    n = gamma.numel()
    scale, shift = gamma.new_empty(n), gamma.new_empty(n)
    check(lib, lib.hf_bn_fold_f32(_p(scale), _p(shift), _p(gamma), _p(beta), _p(mean), _p(var), _p(conv_bias),
                                  float(eps), n, st), "hf_bn_fold_f32")
    return scale, shift


def conv2d(lib, st, x, wt, k, stride=1, in_scale=None, in_shift=None, out_scale=None, bias=None, act=ACT_NONE,
           slope=None, alpha=0.0, residual=None, groups=1, x_shared=True):
    """groups == 1: x [B,cin,H,W], wt [k*k,cin,cout] -> [B,cout,oh,ow].
    groups  > 1: wt [G,k*k,cin,cout], bias/out_scale/slope [G,cout]; x [B,cin,H,W] shared by all
    groups (x_shared) or [G,B,cin,H,W]; returns [G,B,cout,oh,ow]."""
    x = _c(x)
    if groups > 1 and not x_shared:
        g_, b, cin, h, w = x.shape
        if g_ != groups:
            raise ValueError("x must be [groups, B, cin, H, W]")
        x_gstride = b * cin * h * w
    else:
        b, cin, h, w = x.shape
        x_gstride = 0
    cout = wt.shape[-1]
    oh, ow = (h - 1) // stride + 1, (w - 1) // stride + 1
    out = x.new_empty((groups, b, cout, oh, ow) if groups > 1 else (b, cout, oh, ow))
    if residual is not None:
        residual = _c(residual)
        if tuple(residual.shape) != tuple(out.shape):
            raise ValueError(f"residual {tuple(residual.shape)} != output {tuple(out.shape)}")
    n = lib.hf_conv2d_workspace_floats(b, cin, cout, h, w, k, stride, groups)
    ws = x.new_empty((n,)) if n > 0 else None
    code = _launch_profiled(
        lib, 2.0 * cin * cout * k * k * oh * ow * b * groups,
        lambda: lib.hf_conv2d_f32(_p(out), _p(x), _p(_c(wt)), _p(in_scale), _p(in_shift), _p(_c(out_scale)), _p(_c(bias)),
                                  act, _p(_c(slope)), float(alpha), _p(residual), b, cin, cout, h, w, k, stride, groups,
                                  x_gstride, _p(ws), max(n, 0), st))
    check(lib, code, "hf_conv2d_f32")
    return out


def _pow2_ceil(v):
    p = 1
    while p < v:
        p <<= 1
    return p


@functools.lru_cache(maxsize=4096)  # (asked ~700 times per swap with ~60 distinct shapes: host time of a host-bound call sequence)
def conv2d_f16_supported(cin, cout, h, w, k, stride):
    """Shapes hf_conv2d_f16_f32 takes (include/hairfast_hip.h; mirrors launch_enc's tile geometry)."""
    if k != 3 or cin % 16 or cout % 64 or stride not in (1, 2):
        return False
    oh, ow = (h - 1) // stride + 1, (w - 1) // stride + 1
    tw = min(32, _pow2_ceil(ow))
    return tw >= 16 and _pow2_ceil(oh) >= 128 // tw


def conv_split_weights_f16(lib, st, wt):
    """Prepared conv weights [taps,cin,cout] or [G,taps,cin,cout] (taps 9 or 1) -> (hi, lo) fp16 blobs in the layout
    hf_conv2d_f16_f32 / hf_conv1x1_f16_f32 take: per group [taps*cin*cout halves | 16-byte trailer] (hi) and
    taps*cin*cout halves (lo)."""
    wt = _c(wt)
    w4 = wt if wt.ndim == 4 else wt.unsqueeze(0)
    g, taps, cin, cout = w4.shape
    if taps not in (9, 1) or cin % 16:
        raise ValueError(f"f16 MFMA path needs 3x3 or 1x1 weights with cin % 16 == 0; got {tuple(wt.shape)}")
    n = taps * cin * cout
    hi = torch.empty(g * (n + 8), dtype=torch.float16, device=wt.device)
    lo = torch.empty(g * n, dtype=torch.float16, device=wt.device)
    for i in range(g):
        check(lib, lib.hf_conv_split_weights_f16_taps(hi[i * (n + 8):].data_ptr(), lo[i * n:].data_ptr(), w4[i].data_ptr(), cin, cout,
                                                     taps, st), "hf_conv_split_weights_f16_taps")
    return hi, lo


def conv1x1_f16_supported(cin, cout):
    """Shapes hf_conv1x1_f16_f32 takes (include/hairfast_hip.h)."""
    return cin % 32 == 0 and cout % 64 == 0


def conv1x1_f16(lib, st, x, wt_hi, wt_lo, nterms, cout, stride=1, in_scale=None, in_shift=None, out_scale=None, bias=None,
                act=ACT_NONE, slope=None, alpha=0.0, residual=None, groups=1, x_shared=True):
    """hf_conv1x1_f16_f32: a 1x1 conv / Linear layer as a GEMM on the fp16 matrix cores; argument meaning as conv2d().
    x may be a SplitActivation (stride 1, no groups, affine already applied)."""
    pre = isinstance(x, SplitActivation)
    if pre:
        b, cin, h, w = x.shape
        x_gstride = 0
        proto = x.hi
    else:
        x = proto = _c(x)
    if pre:
        pass
    elif groups > 1 and not x_shared:
        g_, b, cin, h, w = x.shape
        if g_ != groups:
            raise ValueError("x must be [groups, B, cin, H, W]")
        x_gstride = b * cin * h * w
    else:
        b, cin, h, w = x.shape
        x_gstride = 0
    oh, ow = (h - 1) // stride + 1, (w - 1) // stride + 1
    out = torch.empty((groups, b, cout, oh, ow) if groups > 1 else (b, cout, oh, ow), dtype=torch.float32, device=proto.device)
    if residual is not None:
        residual = _c(residual)
        if tuple(residual.shape) != tuple(out.shape):
            raise ValueError(f"residual {tuple(residual.shape)} != output {tuple(out.shape)}")
    n = lib.hf_conv1x1_f16_workspace_floats(b, cin, cout, h, w, stride, groups)
    ws = torch.empty((n,), dtype=torch.float32, device=proto.device) if n > 0 else None
    nparts = 2 if nterms == 3 else 1  # algorithmic HBM bytes: the pixels the GEMM reads, the output, a residual, the weights - each once
    n_in = b * (1 if (groups == 1 or x_gstride == 0) else groups)
    nb = (float(n_in) * cin * oh * ow * (2.0 * nparts if pre else 4.0) + float(b) * groups * cout * oh * ow * (4.0 + (4.0 if residual is not None else 0.0))
          + 2.0 * nparts * cin * cout * groups)
    code = _launch_profiled(
        lib, 2.0 * cin * cout * oh * ow * b * groups,
        lambda: lib.hf_conv1x1_f16_f32(_p(out), None if pre else _p(x), _p(x.hi) if pre else None, _p(x.lo) if pre else None,
                                       _p(wt_hi), _p(wt_lo), nterms, _p(_c(in_scale)), _p(_c(in_shift)),
                                       _p(_c(out_scale)), _p(_c(bias)), act, _p(_c(slope)), float(alpha), _p(residual), b, cin, cout,
                                       h, w, stride, groups, x_gstride, _p(ws), max(n, 0), st),
        label="gemm_h", nbytes=nb)
    check(lib, code, "hf_conv1x1_f16_f32")
    return out


def split_activation_f16(lib, st, x, in_scale=None, in_shift=None, want_lo=True):
    """hf_split_activation_f16: x [..., C, H, W] fp32 -> SplitActivation of in_scale*x + in_shift
    (fp16 hi/lo pairs, K-blocked [images, C/8, H, W, 8]); leading dims are flattened into images."""
    x = _c(x)
    c, h, w = x.shape[-3:]
    images = x.numel() // (c * h * w)
    hi = torch.empty((images, c // 8, h, w, 8), dtype=torch.float16, device=x.device)
    lo = torch.empty_like(hi) if want_lo else None
    check(lib, lib.hf_split_activation_f16(_p(hi), _p(lo), _p(x), _p(_c(in_scale)), _p(_c(in_shift)), images, c, h, w, st),
          "hf_split_activation_f16")
    return SplitActivation(hi, lo, None)


def conv2d_f16(lib, st, x, wt_hi, wt_lo, nterms, cout, stride=1, in_scale=None, in_shift=None, out_scale=None, bias=None,
               act=ACT_NONE, slope=None, alpha=0.0, residual=None, groups=1, x_shared=True):
    """hf_conv2d_f16_f32: conv2d(k=3) on the fp16 matrix cores; argument meaning as conv2d().
    x may be a SplitActivation (split_activation_f16 of the input, any in_scale / in_shift already applied):
    [B] images, or [groups*B] images with x_shared=False."""
    pre = isinstance(x, SplitActivation)
    if pre:
        if in_scale is not None or in_shift is not None:
            raise ValueError("a pre-split input carries its affine already")
        n_img, cin, h, w = x.shape
        b = n_img if (groups == 1 or x_shared) else n_img // groups
        x_gstride = 0 if (groups == 1 or x_shared) else 1
        dev = x.hi.device
    else:
        x = _c(x)
        dev = x.device
        if groups > 1 and not x_shared:
            g_, b, cin, h, w = x.shape
            if g_ != groups:
                raise ValueError("x must be [groups, B, cin, H, W]")
            x_gstride = b * cin * h * w
        else:
            b, cin, h, w = x.shape
            x_gstride = 0
    oh, ow = (h - 1) // stride + 1, (w - 1) // stride + 1
    out = torch.empty((groups, b, cout, oh, ow) if groups > 1 else (b, cout, oh, ow), dtype=torch.float32, device=dev)
    if residual is not None:
        residual = _c(residual)
        if tuple(residual.shape) != tuple(out.shape):
            raise ValueError(f"residual {tuple(residual.shape)} != output {tuple(out.shape)}")
    n = lib.hf_conv2d_f16_workspace_floats(b, cin, cout, h, w, stride, groups)
    ws = torch.empty((n,), dtype=torch.float32, device=dev) if n > 0 else None
    # algorithmic HBM bytes (bench.py: traffic / algorithmic): the input once (a shared input of a grouped launch once), the
    # output once, a residual once, the weights once
    nparts = 2 if nterms == 3 else 1
    n_in = b * (1 if (groups == 1 or x_gstride == 0) else groups)
    nb = (float(n_in) * cin * h * w * (2.0 * nparts if pre else 4.0) + float(b) * groups * cout * oh * ow * (4.0 + (4.0 if residual is not None else 0.0))
          + 2.0 * nparts * 9 * cin * cout * groups)
    code = _launch_profiled(
        lib, 2.0 * cin * cout * 9 * oh * ow * b * groups,
        lambda: lib.hf_conv2d_f16_f32(_p(out), None if pre else _p(x), _p(x.hi) if pre else None, _p(x.lo) if pre else None,
                                      _p(wt_hi), _p(wt_lo), nterms, _p(in_scale), _p(in_shift), _p(_c(out_scale)),
                                      _p(_c(bias)), act, _p(_c(slope)), float(alpha), _p(residual), b, cin, cout, h, w, stride,
                                      groups, x_gstride, _p(ws), max(n, 0), st), tag=",pre" if pre else "", nbytes=nb)
    check(lib, code, "hf_conv2d_f16_f32")
    return out


def conv2d_f16_split_supported(lib, b, cin, cout, h, w, stride, nterms=3, pre=False):
    """hf_conv2d_f16_split_f32 takes the launch: groups 1 and no K split spread over the grid (the split output is written by
    the conv kernel's own epilogue; hf_conv2d_f16_split_output_ok).  pre: the input arrives pre-split (SplitActivation)."""
    return cout % 8 == 0 and lib.hf_conv2d_f16_split_output_ok(b, cin, cout, h, w, stride, nterms, 1 if pre else 0) == 1


def conv2d_f16_split(lib, st, x, wt_hi, wt_lo, nterms, cout, stride=1, in_scale=None, in_shift=None, out_scale=None, bias=None,
                     act=ACT_NONE, slope=None, alpha=0.0, residual=None, next_scale=None, next_shift=None, want_f32=False):
    """hf_conv2d_f16_split_f32: conv2d_f16 whose result leaves as a SplitActivation (next_scale * y + next_shift, fp16 hi / lo,
    K-blocked) for the next fp16-core conv; want_f32: also the fp32 tensor -> (SplitActivation, out | None)."""
    pre = isinstance(x, SplitActivation)
    if pre:
        if in_scale is not None or in_shift is not None:
            raise ValueError("a pre-split input carries its affine already")
        b, cin, h, w = x.shape
        dev = x.hi.device
    else:
        x = _c(x)
        dev = x.device
        b, cin, h, w = x.shape
    oh, ow = (h - 1) // stride + 1, (w - 1) // stride + 1
    out = torch.empty((b, cout, oh, ow), dtype=torch.float32, device=dev) if want_f32 else None
    hi = torch.empty((b, cout // 8, oh, ow, 8), dtype=torch.float16, device=dev)
    lo = torch.empty_like(hi) if nterms == 3 else None
    if residual is not None:
        residual = _c(residual)
        if tuple(residual.shape) != (b, cout, oh, ow):
            raise ValueError(f"residual {tuple(residual.shape)} != output {(b, cout, oh, ow)}")
    nparts = 2 if nterms == 3 else 1  # algorithmic HBM bytes: input, split output (+ the fp32 one), residual, weights - each once
    nb = (float(b) * cin * h * w * (2.0 * nparts if pre else 4.0)
          + float(b) * cout * oh * ow * (2.0 * nparts + (4.0 if want_f32 else 0.0) + (4.0 if residual is not None else 0.0))
          + 2.0 * nparts * 9 * cin * cout)
    code = _launch_profiled(
        lib, 2.0 * cin * cout * 9 * oh * ow * b,
        lambda: lib.hf_conv2d_f16_split_f32(_p(out), _p(hi), _p(lo), _p(_c(next_scale)), _p(_c(next_shift)), None if pre else _p(x),
                                            _p(x.hi) if pre else None, _p(x.lo) if pre else None, _p(wt_hi), _p(wt_lo), nterms,
                                            _p(in_scale), _p(in_shift), _p(_c(out_scale)), _p(_c(bias)), act, _p(_c(slope)),
                                            float(alpha), _p(residual), b, cin, cout, h, w, stride, st),
        tag=",pre,split-out" if pre else ",split-out", nbytes=nb)
    check(lib, code, "hf_conv2d_f16_split_f32")
    return SplitActivation(hi, lo, None), out


def plane_mean(lib, st, x):
    x = _c(x)
    b, c, h, w = x.shape
    out = x.new_empty((b, c))
    check(lib, lib.hf_plane_mean_f32(_p(out), _p(x), b * c, h * w, st), "hf_plane_mean_f32")
    return out


def se_gate(lib, st, pooled, fc1, fc2):
    b, c = pooled.shape
    cr = fc1.shape[0]
    gate = pooled.new_empty((b, c))
    check(lib, lib.hf_se_gate_f32(_p(gate), _p(pooled), _p(_c(fc1)), _p(_c(fc2)), b, c, cr, st), "hf_se_gate_f32")
    return gate


def scale_shortcut_add(lib, st, r, gate, shortcut, sc_stride=1):
    r, shortcut = _c(r), _c(shortcut)
    b, c, oh, ow = r.shape
    sh, sw = shortcut.shape[2], shortcut.shape[3]
    out = torch.empty_like(r)
    check(lib, lib.hf_scale_shortcut_add_f32(_p(out), _p(r), _p(gate), _p(shortcut), sc_stride, b, c, oh, ow, sh, sw,
                                             st), "hf_scale_shortcut_add_f32")
    return out


def scale_shortcut_add_split(lib, st, r, gate, shortcut, sc_stride=1, next_scale=None, next_shift=None, want_lo=True):
    """hf_scale_shortcut_add_split_f16: scale_shortcut_add whose result also leaves as the SplitActivation of
    next_scale * out + next_shift (the next unit's first conv input) -> (out fp32, SplitActivation)."""
    r, shortcut = _c(r), _c(shortcut)
    b, c, oh, ow = r.shape
    sh, sw = shortcut.shape[2], shortcut.shape[3]
    out = torch.empty_like(r)
    hi = torch.empty((b, c // 8, oh, ow, 8), dtype=torch.float16, device=r.device)
    lo = torch.empty_like(hi) if want_lo else None
    check(lib, lib.hf_scale_shortcut_add_split_f16(_p(out), _p(hi), _p(lo), _p(_c(next_scale)), _p(_c(next_shift)), _p(r), _p(gate),
                                                   _p(shortcut), sc_stride, b, c, oh, ow, sh, sw, st), "hf_scale_shortcut_add_split_f16")
    return out, SplitActivation(hi, lo, None)


def upsample_bilinear_add(lib, st, x, y):
    x, y = _c(x), _c(y)
    b, c, h, w = x.shape
    oh, ow = y.shape[2], y.shape[3]
    out = torch.empty_like(y)
    check(lib, lib.hf_upsample_bilinear_add_f32(_p(out), _p(x), _p(y), b * c, h, w, oh, ow, st),
          "hf_upsample_bilinear_add_f32")
    return out


def adaptive_avgpool_into(lib, st, out, x, c_off):
    """AdaptiveAvgPool2d(out.shape[2:]) of x written into out[:, c_off:c_off+C]."""
    x = _c(x)
    b, c, h, w = x.shape
    check(lib, lib.hf_adaptive_avgpool_f32(_p(out), _p(x), b, c, h, w, out.shape[2], out.shape[3], out.shape[1], c_off,
                                           st), "hf_adaptive_avgpool_f32")


def downscale2x(lib, st, x):
    x = _c(x)
    b, c, h, w = x.shape
    out = x.new_empty((b, c, h // 2, w // 2))
    check(lib, lib.hf_downscale2x_f32(_p(out), _p(x), b * c, h, w, st), "hf_downscale2x_f32")
    return out


def linear(lib, st, x, weight, bias, scale=1.0):
    """x [B,in] (row stride may exceed in), weight [out,in] -> [B,out]; B <= 8 per launch."""
    if x.stride(-1) != 1:
        x = x.contiguous()
    b, k = x.shape
    weight = _c(weight)
    n = weight.shape[0]
    out = x.new_empty((b, n))
    check(lib, lib.hf_linear_f32(_p(out), _p(x), x.stride(0) if b > 1 else k, _p(weight), _p(_c(bias)), b, k, n, float(scale), st),
          "hf_linear_f32")
    return out


def equal_linear(lib, st, x, weight, bias, lr_mul=1.0, fused_lrelu=False, alpha=0.2, act_scale=SQRT2):
    """EqualLinear.forward (model.py:153-163) in one launch per <= 8 rows: x [..., in] -> [..., out]."""
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    if x2.dtype != torch.float32:
        raise TypeError("hairfastgan_amd kernels are fp32")
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    b, k = x2.shape
    weight = _c(weight)
    n = weight.shape[0]
    out = x2.new_empty((b, n))
    check(lib, lib.hf_equal_linear_f32(_p(out), _p(x2), x2.stride(0) if b > 1 else k, _p(weight), _p(_c(bias)), b, k, n,
                                       float(lr_mul), 1 if fused_lrelu else 0, float(alpha), float(act_scale), st),
          "hf_equal_linear_f32")
    return out.reshape(*lead, n)


def pixel_norm(lib, st, x):
    """PixelNorm over dim 1 of [B, dim] (model.py:16-21)."""
    x = _c(x)
    if x.ndim != 2:
        raise ValueError("pixel_norm expects [B, dim] (the mapping network's z)")
    out = torch.empty_like(x)
    check(lib, lib.hf_pixel_norm_f32(_p(out), _p(x), x.shape[0], x.shape[1], st), "hf_pixel_norm_f32")
    return out


def bicubic_down(lib, st, x, k1d, factor):
    """BicubicDownSample.forward (utils/bicubic.py): x [B,C,H,W] -> [B,C,H/factor,W/factor]."""
    x, k1d = _c(x), _c(k1d)
    b, c, h, w = x.shape
    out = x.new_empty((b, c, h // factor, w // factor))
    check(lib, lib.hf_bicubic_down_f32(_p(out), _p(x), _p(k1d), b * c, h, w, factor, st), "hf_bicubic_down_f32")
    return out


def dilate_erode(lib, st, mask, radius):
    """DilateErosion.mask on a binary float mask [..., H, W] -> (dilated, eroded)."""
    mask = _c(mask)
    h, w = mask.shape[-2:]
    dil, ero = torch.empty_like(mask), torch.empty_like(mask)
    check(lib, lib.hf_dilate_erode_f32(_p(dil), _p(ero), _p(mask), mask.numel() // (h * w), h, w, radius, st), "hf_dilate_erode_f32")
    return dil, ero


def stem_prepare(w):
    """[cout,3,7,7] conv weight -> (w_hi, w_lo, w_unscale) of hf_stem7x7s2_f16_f32: fp16 (hi, lo) parts of W * 2^k in
    [cout/64][22 K groups][64][8] - group (ci, ky), eight halves = kx 0..6 + a zero; the 22nd group zero - with 2^k placing
    max|W| in [2^13, 2^14) (the lo parts of small weights would otherwise be fp16 subnormals, csrc/convh.hip split_weights)."""
    cout = w.shape[0]
    if tuple(w.shape[1:]) != (3, 7, 7) or cout % 64:
        raise ValueError(f"stem weight must be [64n,3,7,7]; got {tuple(w.shape)}")
    w = w.detach().float()
    amax = w.abs().max().clamp_min(1e-30)
    k = torch.floor(13.0 - torch.log2(amax))
    scale = torch.exp2(k)
    ws = F.pad(w * scale, (0, 1)).reshape(cout, 21, 8)                           # kx padded to 8
    ws = torch.cat([ws, ws.new_zeros(cout, 1, 8)], 1)                            # the zero group
    ws = ws.reshape(cout // 64, 64, 22, 8).permute(0, 2, 1, 3).contiguous()      # [tiles][22][64][8]
    hi = ws.half()
    lo = (ws - hi.float()).half()
    return hi.contiguous(), lo.contiguous(), torch.exp2(-k).reshape(1).float().contiguous()


def stem7x7s2(lib, st, x, w3, out_scale=None, bias=None, alpha=0.0, pool=True):
    """hf_stem7x7s2_f16_f32: 7x7 stride-2 conv of a 3-channel image + per-channel affine + leaky ReLU (+ 3x3/2 max pool)."""
    x = _c(x)
    b, cin, h, w = x.shape
    hi, lo, unscale = w3
    cout = hi.shape[0] * 64
    if cin != 3:
        raise ValueError("stem input must have 3 channels")
    oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    if pool:
        oh, ow = (oh - 1) // 2 + 1, (ow - 1) // 2 + 1
    out = x.new_empty((b, cout, oh, ow))
    code = _launch_profiled(
        lib, 2.0 * 3 * 49 * cout * ((h - 1) // 2 + 1) * ((w - 1) // 2 + 1) * b,
        lambda: lib.hf_stem7x7s2_f16_f32(_p(out), _p(x), _p(hi), _p(lo), _p(unscale), _p(_c(out_scale)), _p(_c(bias)), float(alpha),
                                         b, h, w, cout, 1 if pool else 0, st),
        label="stem 7x7/2 (+pool)")
    check(lib, code, "hf_stem7x7s2_f16_f32")
    return out


def maxpool3x3s2(lib, st, x):
    x = _c(x)
    b, c, h, w = x.shape
    out = x.new_empty((b, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1))
    check(lib, lib.hf_maxpool3x3s2_f32(_p(out), _p(x), b * c, h, w, st), "hf_maxpool3x3s2_f32")
    return out


def gate(lib, st, x, logit, add_plane=None, add_bcast=None, plus_one=0.0):
    """x * (sigmoid(logit[b,c]) + plus_one) + add_plane + add_bcast[b,c] (hf_gate_f32)."""
    x = _c(x)
    b, c, h, w = x.shape
    out = torch.empty_like(x)
    check(lib, lib.hf_gate_f32(_p(out), _p(x), _p(_c(logit)), _p(_c(add_plane)), _p(_c(add_bcast)), float(plus_one), b * c, h * w, st),
          "hf_gate_f32")
    return out


def upsample_nearest(lib, st, x, oh, ow):
    x = _c(x)
    b, c, h, w = x.shape
    out = x.new_empty((b, c, oh, ow))
    check(lib, lib.hf_upsample_nearest_f32(_p(out), _p(x), b * c, h, w, oh, ow, st), "hf_upsample_nearest_f32")
    return out


def parsing_mask(lib, st, logits, remap, full_hw, out_hw):
    """hf_parsing_mask_i64: class logits [B,C,h,w] -> int64 mask [B,1,oh,ow] (bilinear up to full_hw, nearest resize
    to out_hw, first-max argmax, label permutation remap int32 [C])."""
    logits = _c(logits)
    b, c, h, w = logits.shape
    out = torch.empty((b, 1, out_hw[0], out_hw[1]), dtype=torch.int64, device=logits.device)
    check(lib, lib.hf_parsing_mask_i64(_p(out), _p(logits), _p(remap), b, c, h, w, full_hw[0], full_hw[1], out_hw[0], out_hw[1], st),
          "hf_parsing_mask_i64")
    return out


def layernorm(lib, st, x, dim, gamma=None, beta=None, eps=1e-5, lrelu=False, alpha=0.01, groups=1):
    """F.layer_norm over the trailing `dim` elements (x viewed as [rows, dim]), optional affine + LeakyReLU.
    groups > 1 (hf_layernorm_grouped_f32): gamma / beta [groups, dim], row r of the [rows, dim] view takes group r % groups."""
    x = _c(x)
    if x.numel() % dim:
        raise ValueError("dim must divide the tensor")
    out = torch.empty_like(x)
    if groups > 1:
        rows = x.numel() // dim
        if rows % groups or gamma is None or tuple(gamma.shape) != (groups, dim) or (beta is not None and tuple(beta.shape) != (groups, dim)):
            raise ValueError("grouped layernorm: rows % groups == 0 and gamma / beta [groups, dim]")
        check(lib, lib.hf_layernorm_grouped_f32(_p(out), _p(x), _p(_c(gamma)), _p(_c(beta)), rows, dim, groups, float(eps),
                                                1 if lrelu else 0, float(alpha), st), "hf_layernorm_grouped_f32")
        return out
    check(lib, lib.hf_layernorm_f32(_p(out), _p(x), _p(_c(gamma)), _p(_c(beta)), x.numel() // dim, dim, float(eps),
                                    1 if lrelu else 0, float(alpha), st), "hf_layernorm_f32")
    return out


def sample_layernorm(lib, st, x, gamma=None, beta=None, eps=1e-5, slope=1.0, channels=None):
    """hf_sample_layernorm_f32: per-sample LayerNorm over C*H*W with the unbiased std and eps added to it, per-channel
    affine, LeakyReLU(slope) - the CtrlHair shape adaptor's norm.  channels < x.shape[1]: only the first `channels`
    planes of every sample are normalised and returned (x came out of a conv with a padded channel count)."""
    x = _c(x)
    b, c_all = x.shape[0], x.shape[1]
    c = c_all if channels is None else channels
    hw = x.numel() // (b * c_all)
    out = x.new_empty((b, c) + tuple(x.shape[2:]))
    n = lib.hf_sample_layernorm_workspace_floats(b, c, hw)
    ws = x.new_empty((max(n, 1),))
    check(lib, lib.hf_sample_layernorm_f32(_p(out), _p(x), _p(_c(gamma)), _p(_c(beta)), b, c, hw, c_all * hw, float(eps),
                                           float(slope), _p(ws), n, st), "hf_sample_layernorm_f32")
    return out


def modulate(lib, st, x, gamma, beta, lrelu=False, alpha=0.01):
    x, gamma, beta = _c(x), _c(gamma), _c(beta)
    if gamma.shape != x.shape or beta.shape != x.shape:
        raise ValueError("gamma / beta must have x's shape")
    out = torch.empty_like(x)
    check(lib, lib.hf_modulate_f32(_p(out), _p(x), _p(gamma), _p(beta), x.numel(), 1 if lrelu else 0, float(alpha), st),
          "hf_modulate_f32")
    return out


def pixel_norm_dim1(lib, st, x):
    x = _c(x)
    b, layers, d = x.shape
    out = torch.empty_like(x)
    check(lib, lib.hf_pixel_norm_dim1_f32(_p(out), _p(x), b, layers, d, st), "hf_pixel_norm_dim1_f32")
    return out


def axpby(lib, st, a, alpha, bvec, beta=1.0):
    """alpha * a + beta * bvec, bvec broadcast periodically over a (hf_axpby_bcast_f32)."""
    a, bvec = _c(a), _c(bvec)
    out = torch.empty_like(a)
    check(lib, lib.hf_axpby_bcast_f32(_p(out), _p(a), float(alpha), _p(bvec), float(beta), a.numel(), bvec.numel(), st),
          "hf_axpby_bcast_f32")
    return out


def add_bcast(lib, st, a, bvec):
    a, bvec = _c(a), _c(bvec)
    out = torch.empty_like(a)
    check(lib, lib.hf_add_bcast_f32(_p(out), _p(a), _p(bvec), a.numel(), bvec.numel(), st), "hf_add_bcast_f32")
    return out


# ----------------------------------------------------------------------------------------
# SEAN (csrc/sean.hip)
# ----------------------------------------------------------------------------------------
def label_conv3x3(lib, st, labels, table, bias, channels, batch=None, cols_per_sample=0, group=1, relu=False):
    """hf_label_conv3x3_f32: 3x3 conv of a per-label-constant input as nine table lookups per output.
    labels int32 [B/group, H, W]; table [9*channels, table_cols] -> [B, channels, H, W]."""
    if labels.dtype != torch.int32 or not labels.is_contiguous():
        raise TypeError("labels must be contiguous int32 [B/group, H, W]")
    table = _c(table)
    nb, h, w = labels.shape
    b = nb * group if batch is None else batch
    if table.shape[0] != 9 * channels or (cols_per_sample and table.shape[1] < b * cols_per_sample):
        raise ValueError(f"table {tuple(table.shape)} does not fit {channels} channels / {b} samples")
    out = table.new_empty((b, channels, h, w))
    # interior fast path (one lookup of the tap sum instead of nine) where planes are large enough to have interiors
    tsum = table.new_empty((channels, table.shape[1])) if h * w >= 1024 else None
    check(lib, lib.hf_label_conv3x3_f32(_p(out), _p(labels), _p(table), _p(_c(bias)), b, channels, h, w, table.shape[1],
                                        cols_per_sample, group, 1 if relu else 0, _p(tsum), st), "hf_label_conv3x3_f32")
    return out


def ace_modulate(lib, st, x, noise, noise_var, bn_scale, bn_shift, avg, sp, blend, group=1, slope=1.0, x_up=False):
    """hf_ace_modulate_f32: the tail of ACE.forward; x [B,C,H,W], noise [B,H,W] | None, avg [B,2C,H,W] | None,
    sp [B/group,2C,H,W], blend = device tensor (blending_gamma, blending_beta).  x_up: x is [B,C,H/2,W/2] and its nearest
    x2 up-sampling is read in place."""
    x, sp = _c(x), _c(sp)
    b, c = x.shape[:2]
    h, w = sp.shape[-2:]
    if tuple(x.shape[-2:]) != ((h // 2, w // 2) if x_up else (h, w)):
        raise ValueError(f"x {tuple(x.shape)} does not match the {h} x {w} planes of sp (x_up={x_up})")
    if tuple(sp.shape) != (b // group, 2 * c, h, w) or (avg is not None and tuple(avg.shape) != (b, 2 * c, h, w)):
        raise ValueError("sp / avg must be [B/group | B, 2C, H, W]")
    if noise is not None and noise.numel() != b * h * w:
        raise ValueError("noise must be [B, H, W]")
    out = x.new_empty((b, c, h, w))
    check(lib, lib.hf_ace_modulate_f32(_p(out), _p(x), _p(_c(noise)), _p(_c(noise_var)), _p(_c(bn_scale)), _p(_c(bn_shift)),
                                       _p(_c(avg)), _p(sp), _p(_c(blend)), b, c, h * w, group, float(slope), w if x_up else 0, st),
          "hf_ace_modulate_f32")
    return out


def ace_modulate_table(lib, st, x, noise, noise_var, bn_scale, bn_shift, labels, table, avg_bias, sp, blend, group=1, slope=1.0,
                       x_up=False):
    """hf_ace_modulate_table_f32: label_conv3x3(labels, table, avg_bias, 2C, batch=B, cols_per_sample=19, group) fused into
    ace_modulate (the [B,2C,H,W] avg planes never exist).  table [9*2C, >= B*19]; labels int32 [B/group,H,W]."""
    x, sp, table = _c(x), _c(sp), _c(table)
    b, c = x.shape[:2]
    h, w = sp.shape[-2:]
    if tuple(x.shape[-2:]) != ((h // 2, w // 2) if x_up else (h, w)):
        raise ValueError(f"x {tuple(x.shape)} does not match the {h} x {w} planes of sp (x_up={x_up})")
    if labels.dtype != torch.int32 or not labels.is_contiguous() or tuple(labels.shape) != (b // group, h, w):
        raise TypeError("labels must be contiguous int32 [B/group, H, W]")
    if tuple(sp.shape) != (b // group, 2 * c, h, w) or table.shape[0] != 9 * 2 * c or table.shape[1] < b * 19:
        raise ValueError("sp must be [B/group, 2C, H, W] and table [9*2C, >= 19*B]")
    if noise is not None and noise.numel() != b * h * w:
        raise ValueError("noise must be [B, H, W]")
    out = x.new_empty((b, c, h, w))
    check(lib, lib.hf_ace_modulate_table_f32(_p(out), _p(x), _p(_c(noise)), _p(_c(noise_var)), _p(_c(bn_scale)), _p(_c(bn_shift)),
                                             _p(labels), _p(table), _p(_c(avg_bias)), _p(sp), _p(_c(blend)), b, c, h, w,
                                             table.shape[1], 19, group, float(slope), 1 if h * w >= 1024 else 0, 1 if x_up else 0, st),
          "hf_ace_modulate_table_f32")
    return out


def ace_modulate_table_supported(h, w):
    return w % 4 == 0 and h * w >= 1024


def region_mean(lib, st, x, labels, crop=0, act_tanh=False):
    """hf_region_mean_f32: per-label mean of (tanh of) x [B,C,H+2*crop,W+2*crop] over its interior, labels int32
    [B,H,W] -> [B,19,C]."""
    x = _c(x)
    b, c, hp, wp = x.shape
    h, w = hp - 2 * crop, wp - 2 * crop
    if labels.dtype != torch.int32 or tuple(labels.shape) != (b, h, w) or not labels.is_contiguous():
        raise TypeError("labels must be contiguous int32 [B, H, W] of the cropped size")
    out = x.new_empty((b, 19, c))
    base = x.data_ptr() + 4 * (crop * wp + crop)
    check(lib, lib.hf_region_mean_f32(_p(out), base, _p(labels), b, c, h, w, 19, c * hp * wp, hp * wp, wp, 1 if act_tanh else 0, st),
          "hf_region_mean_f32")
    return out


def tanh(lib, st, x):
    x = _c(x)
    out = torch.empty_like(x)
    check(lib, lib.hf_tanh_f32(_p(out), _p(x), x.numel(), st), "hf_tanh_f32")
    return out


# ----------------------------------------------------------------------------------------
# CLIP ViT image tower (csrc/vit.hip): feature-major activations [C, T]
# ----------------------------------------------------------------------------------------
def channel_layernorm(lib, st, x, gamma, beta, eps=1e-5):
    """LayerNorm over the feature axis of x [C, ...tokens] or [1, C, ...tokens] (the NCHW view the 1x1-conv GEMMs use)."""
    x = _c(x)
    c = x.shape[1] if (x.ndim == 4 and x.shape[0] == 1) else x.shape[0]
    out = torch.empty_like(x)
    check(lib, lib.hf_channel_layernorm_f32(_p(out), _p(x), _p(_c(gamma)), _p(_c(beta)), c, x.numel() // c, float(eps), st),
          "hf_channel_layernorm_f32")
    return out


def mha_small(lib, st, qkv, images, seq, heads):
    """qkv [3E, images*seq] (or [1, 3E, images, seq]) feature-major -> attention output [E, images*seq] (same form)."""
    qkv = _c(qkv)
    lead = 1 if (qkv.ndim == 4 and qkv.shape[0] == 1) else 0
    e = qkv.shape[lead] // 3
    if qkv.numel() != 3 * e * images * seq:
        raise ValueError("qkv must be [3E, images*seq]")
    out = qkv.new_empty(tuple(qkv.shape[:lead]) + (e,) + tuple(qkv.shape[lead + 1:]))
    check(lib, lib.hf_mha_small_f32(_p(out), _p(qkv), images, seq, heads, e // heads, st), "hf_mha_small_f32")
    return out


def quick_gelu(lib, st, x):
    x = _c(x)
    out = torch.empty_like(x)
    check(lib, lib.hf_quick_gelu_f32(_p(out), _p(x), x.numel(), st), "hf_quick_gelu_f32")
    return out
