"""`torch.ops.hairfast.*` - the entry points of libhairfast_hip.so registered as torch custom operators.

The reference loads its two native plugins as pybind modules (`fused`, `upfirdn2d`: op/fused_act.py:10-17,
op/upfirdn2d.py:10-17) and calls `fused.fused_bias_act(...)` / `upfirdn2d_op.upfirdn2d(...)`
(fused_bias_act.cpp:11-17, upfirdn2d.cpp:12-19).  Here the same operators - and the fused ones this backend
adds - are registered with the torch dispatcher (`torch.library`): a schema, an implementation for the
`CUDA` dispatch key (= HIP on ROCm) that allocates the output through the caching allocator and calls the C
ABI on torch's current stream, and a fake (meta) implementation for shape propagation, so that they are
visible to `torch.ops`, FakeTensor tracing and `torch.compile` graphs.  There is deliberately no `CPU`
implementation: the dispatcher itself reports a missing kernel (the CPU restatement lives in oracle/ and is
test infrastructure).

    torch.ops.hairfast.upfirdn2d(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)
    torch.ops.hairfast.fused_bias_act(input, bias, alpha, scale)
    torch.ops.hairfast.noise_bias_act(x, noise, noise_w, bias, alpha, scale)
    torch.ops.hairfast.modulated_conv3x3(x, wt, s, d, noise, noise_w, bias, alpha, scale)
    torch.ops.hairfast.modulated_conv3x3_up(x, wt, s, d, blur_kernel, noise, noise_w, bias, alpha, scale)
    torch.ops.hairfast.to_rgb(x, wt, s, bias, skip, up_kernel)
    torch.ops.hairfast.conv2d(x, wt, k, stride, in_scale, in_shift, out_scale, bias, act, slope, alpha, residual)

The public functions of `hairfastgan_amd.stylegan2.op` go through these operators.  The generator's fused
layer chain (`Generator.forward`) keeps calling the C ABI through `_marshal` directly: its launches are
5-10 us apart at batch 1 and a dispatcher round trip per launch would be the bottleneck there.
"""
import torch

from . import _marshal as M
from ._runtime import lib, stream

_LIB = torch.library.Library("hairfast", "DEF")

_SCHEMAS = {
    "upfirdn2d": "(Tensor input, Tensor kernel, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, "
                 "int pad_y0, int pad_y1) -> Tensor",
    "fused_bias_act": "(Tensor input, Tensor? bias, float alpha, float scale) -> Tensor",
    "noise_bias_act": "(Tensor x, Tensor? noise, Tensor? noise_w, Tensor? bias, float alpha, float scale) -> Tensor",
    "modulated_conv3x3": "(Tensor x, Tensor wt, Tensor? s, Tensor? d, Tensor? noise, Tensor? noise_w, Tensor? bias, float alpha, "
                         "float scale) -> Tensor",
    "modulated_conv3x3_up": "(Tensor x, Tensor wt, Tensor? s, Tensor? d, Tensor blur_kernel, Tensor? noise, Tensor? noise_w, "
                            "Tensor? bias, float alpha, float scale) -> Tensor",
    "to_rgb": "(Tensor x, Tensor wt, Tensor? s, Tensor? bias, Tensor? skip, Tensor? up_kernel) -> Tensor",
    "conv2d": "(Tensor x, Tensor wt, int k, int stride, Tensor? in_scale, Tensor? in_shift, Tensor? out_scale, Tensor? bias, "
              "int act, Tensor? slope, float alpha, Tensor? residual) -> Tensor",
}
for _name, _schema in _SCHEMAS.items():
    _LIB.define(_name + _schema)


# ---- HIP implementations (dispatch key CUDA) ------------------------------------------------------
def _upfirdn2d(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    return M.upfirdn2d(lib(), stream(), input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)


def _fused_bias_act(input, bias, alpha, scale):
    return M.fused_bias_act(lib(), stream(), input, bias, alpha, scale)


def _noise_bias_act(x, noise, noise_w, bias, alpha, scale):
    return M.noise_bias_act(lib(), stream(), x, noise, noise_w, bias, alpha, scale)


def _modulated_conv3x3(x, wt, s, d, noise, noise_w, bias, alpha, scale):
    return M.modconv3x3(lib(), stream(), x, wt, s, d, noise, noise_w, bias, alpha, scale)


def _modulated_conv3x3_up(x, wt, s, d, blur_kernel, noise, noise_w, bias, alpha, scale):
    return M.modconv3x3_up(lib(), stream(), x, wt, s, d, blur_kernel, noise, noise_w, bias, alpha, scale)


def _to_rgb(x, wt, s, bias, skip, up_kernel):
    return M.torgb(lib(), stream(), x, wt, s, bias, skip, up_kernel)


def _conv2d(x, wt, k, stride, in_scale, in_shift, out_scale, bias, act, slope, alpha, residual):
    return M.conv2d(lib(), stream(), x, wt, k, stride, in_scale=in_scale, in_shift=in_shift, out_scale=out_scale, bias=bias,
                    act=act, slope=slope, alpha=alpha, residual=residual)


for _name, _fn in (("upfirdn2d", _upfirdn2d), ("fused_bias_act", _fused_bias_act), ("noise_bias_act", _noise_bias_act),
                   ("modulated_conv3x3", _modulated_conv3x3), ("modulated_conv3x3_up", _modulated_conv3x3_up),
                   ("to_rgb", _to_rgb), ("conv2d", _conv2d)):
    _LIB.impl(_name, _fn, "CUDA")


# ---- fake (meta) implementations: output shapes only ----------------------------------------------------
@torch.library.register_fake("hairfast::upfirdn2d")
def _(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    n, c, h, w = input.shape
    kh, kw = kernel.shape
    return input.new_empty((n, c, (h * up_y + pad_y0 + pad_y1 - kh) // down_y + 1, (w * up_x + pad_x0 + pad_x1 - kw) // down_x + 1))


@torch.library.register_fake("hairfast::fused_bias_act")
def _(input, bias, alpha, scale):
    return torch.empty_like(input)


@torch.library.register_fake("hairfast::noise_bias_act")
def _(x, noise, noise_w, bias, alpha, scale):
    return torch.empty_like(x)


@torch.library.register_fake("hairfast::modulated_conv3x3")
def _(x, wt, s, d, noise, noise_w, bias, alpha, scale):
    return x.new_empty((x.shape[0], wt.shape[2], x.shape[2], x.shape[3]))


@torch.library.register_fake("hairfast::modulated_conv3x3_up")
def _(x, wt, s, d, blur_kernel, noise, noise_w, bias, alpha, scale):
    return x.new_empty((x.shape[0], wt.shape[2], 2 * x.shape[2], 2 * x.shape[3]))


@torch.library.register_fake("hairfast::to_rgb")
def _(x, wt, s, bias, skip, up_kernel):
    return x.new_empty((x.shape[0], 3, x.shape[2], x.shape[3]))


@torch.library.register_fake("hairfast::conv2d")
def _(x, wt, k, stride, in_scale, in_shift, out_scale, bias, act, slope, alpha, residual):
    return x.new_empty((x.shape[0], wt.shape[-1], (x.shape[2] - 1) // stride + 1, (x.shape[3] - 1) // stride + 1))
