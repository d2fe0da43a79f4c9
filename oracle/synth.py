"""Deterministic closed-form parameter / input fill (TEST INFRASTRUCTURE).

No pretrained checkpoints exist in the build or on the GPU box and a 121 MB
state dict cannot be committed, so both the imported reference (in
``make_golden.py``) and the HIP path (tests, bench) fill every tensor with the
same pure function of ``(key, shape)``.  The function is integer arithmetic
(splitmix64 finaliser over the flat index, salted with crc32(key)) mapped to
24-bit uniform floats, so it is bit-identical on any machine / numpy version.

Crucially the parameters the reference zero-initialises (``*.noise.weight``,
``*.activate.bias``, ``to_rgb*.bias``; models/stylegan2/model.py:286,
op/fused_act.py:77, model.py:354) get NON-zero values here, otherwise a kernel
that ignores noise or bias would still pass (determinism traps).
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def uniform01(key, shape):
    """Uniform floats in [0, 1) with 24 random bits, fp32-exact."""
    n = int(np.prod(shape)) if len(shape) else 1
    salt = np.uint64(zlib.crc32(key.encode()) & 0xFFFFFFFF) << np.uint64(32)
    with np.errstate(over="ignore"):
        h = _mix64(np.arange(n, dtype=np.uint64) ^ salt)
    u = (h >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))
    return u.reshape(shape)


def unit_uniform(key, shape):
    """Zero mean, unit variance, bounded: uniform on [-sqrt3, sqrt3)."""
    return (uniform01(key, shape) * np.float32(2.0) - np.float32(1.0)) * np.float32(3.0 ** 0.5)


def pseudo_normal(key, shape):
    """Approximately N(0,1) (Irwin-Hall of 4 uniforms), bounded by 2*sqrt3."""
    acc = np.zeros(shape, dtype=np.float32)
    for i in range(4):
        acc += uniform01(f"{key}#{i}", shape)
    return (acc - np.float32(2.0)) * np.float32(3.0 ** 0.5)


def fill_value(key, shape):
    """Value for a state-dict entry of the StyleGAN2 generator (and encoders)."""
    shape = tuple(shape)
    leaf = key.rsplit(".", 1)[-1]
    if key.endswith("blur.kernel") or key.endswith("upsample.kernel"):
        k = np.array([1.0, 3.0, 3.0, 1.0], dtype=np.float32)
        k2 = np.outer(k, k)
        return (k2 / k2.sum() * np.float32(4.0)).astype(np.float32)  # make_kernel * factor**2
    if key.startswith("noises."):
        return pseudo_normal(key, shape)
    if key.endswith("noise.weight"):
        return (np.float32(0.05) + np.float32(0.1) * uniform01(key, shape)).astype(np.float32)
    if key.endswith("activate.bias"):
        return (np.float32(0.2) * unit_uniform(key, shape)).astype(np.float32)
    if key.endswith("modulation.bias"):
        return (np.float32(1.0) + np.float32(0.1) * unit_uniform(key, shape)).astype(np.float32)
    if key.startswith("to_rgb") and leaf == "bias":
        return (np.float32(0.1) * unit_uniform(key, shape)).astype(np.float32)
    if "running_var" in key:
        return (np.float32(0.5) + uniform01(key, shape)).astype(np.float32)
    if "num_batches_tracked" in key:
        return np.zeros(shape, dtype=np.int64)
    if leaf == "bias" or "running_mean" in key:
        return (np.float32(0.1) * unit_uniform(key, shape)).astype(np.float32)
    # ---- encoder parameters (plain Conv2d / BatchNorm2d / PReLU / Linear, no runtime scale) ----
    if leaf == "weight" and len(shape) == 4:  # conv [cout, cin, k, k]: keep activations O(1)
        fan_in = shape[1] * shape[2] * shape[3]
        return (unit_uniform(key, shape) * np.float32(0.7 / np.sqrt(fan_in))).astype(np.float32)
    if leaf == "weight" and len(shape) == 1:  # BatchNorm gamma / PReLU slope: positive, < 1
        return (np.float32(0.1) + np.float32(0.8) * uniform01(key, shape)).astype(np.float32)
    if leaf == "weight" and len(shape) == 2 and _is_plain_linear(key):  # nn.Linear (FS encoder heads)
        return (unit_uniform(key, shape) * np.float32(1.0 / np.sqrt(shape[1]))).astype(np.float32)
    return unit_uniform(key, shape)


def _is_plain_linear(key):
    """`styles.<i>.weight` of the FeatureStyle encoder (nn.Linear, no equalised-lr scale);
    the e4e heads are `styles.<i>.linear.weight` (EqualLinear, scaled at run time)."""
    parts = key.split(".")
    return len(parts) >= 3 and parts[-3] == "styles" and parts[-2].isdigit()


def fill_state_dict(shapes):
    """shapes: mapping key -> shape.  Returns key -> np.ndarray."""
    return {k: fill_value(k, s) for k, s in shapes.items()}


def latent_wplus(batch, seed_key="wplus", n_latent=18, dim=512):
    return pseudo_normal(f"{seed_key}/{batch}", (batch, n_latent, dim))


def noise_maps(seed_key="noise", log_size=10, batch=1):
    """Explicit per-layer noise, shapes as make_noise (model.py:455-464)."""
    out = [pseudo_normal(f"{seed_key}/0", (batch, 1, 4, 4))]
    li = 1
    for i in range(3, log_size + 1):
        for _ in range(2):
            out.append(pseudo_normal(f"{seed_key}/{li}", (batch, 1, 2 ** i, 2 ** i)))
            li += 1
    return out
