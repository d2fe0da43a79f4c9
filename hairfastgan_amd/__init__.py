"""hairfastgan_amd - MI355X (gfx950) native backend for HairFastGAN's generator hot path.

Hand-written HIP kernels behind a C ABI (include/hairfast_hip.h, built in-tree as
hairfastgan_amd/csrc/libhairfast_hip.so) with a host-side mirror of the reference's
operator/module interface:

  hairfastgan_amd.stylegan2.op      <->  models/stylegan2/op   (FusedLeakyReLU, fused_leaky_relu, upfirdn2d)
  hairfastgan_amd.stylegan2.model   <->  models/stylegan2/model.py (Generator, StyledConv, ToRGB, ...)
  hairfastgan_amd.net.Net           <->  models/Net.py:20-46 (owner of `.generator`, `.latent_avg`)

There is no CPU or eager-PyTorch fallback: ops raise on CPU tensors and importing
them without the built library raises.
"""
__version__ = "0.1.0"
