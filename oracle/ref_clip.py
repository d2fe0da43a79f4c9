"""TEST INFRASTRUCTURE - CPU restatement of the CLIP ViT-B/32 image tower.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (hairfastgan_amd/) never does.

PARITY UNPINNED.  The tower is `clip_model.encode_image` of models/Encoders.py:75-94, where `clip_model` comes from
`clip.load("ViT-B/32")` of the reference's un-vendored dependency `clip @ git+https://github.com/openai/CLIP@a1d0717`
(requirements.txt:6); its source is not under /root/reference and the package is not installed in the build container,
so this file restates the PUBLISHED architecture of that revision's clip/model.py:

  VisionTransformer(input_resolution=224, patch_size=32, width=768, layers=12, heads=12, output_dim=512)
    conv1 (3 -> 768, kernel 32, stride 32, no bias) -> [B,768,7,7] -> tokens [B,49,768]; class_embedding prepended;
    + positional_embedding [50,768]; ln_pre; 12 x ResidualAttentionBlock; ln_post on the class token; @ proj [768,512]
  ResidualAttentionBlock: x = x + attn(ln_1(x)); x = x + c_proj(QuickGELU(c_fc(ln_2(x))))
    attn = nn.MultiheadAttention(768, 12) (in_proj_weight [2304,768], in_proj_bias, out_proj), no mask for the vision tower
  QuickGELU: x * sigmoid(1.702 x);  LayerNorm: nn.LayerNorm evaluated in fp32 (eps 1e-5)

and is anchored on what IS available here: the call site (Encoders.py:91-94: AdaptiveAvgPool2d(224) -> *0.5+0.5 ->
CLIP's normalisation -> encode_image) - pinned with the real ClipBlendingModel in make_golden.py - and torch's own
nn.MultiheadAttention / nn.LayerNorm, against which tests/test_sim_clip.py checks this restatement's attention and
normalisation.  State-dict keys are those of the OpenAI model's `visual.*` entries.  The reference runs the tower in
fp16 on the GPU (clip.load(device="cuda") converts the weights); this restatement - like the product - computes in fp32.
"""
import torch
import torch.nn.functional as F

WIDTH, LAYERS, HEADS, PATCH, RES, OUT_DIM = 768, 12, 12, 32, 224, 512


def clip_visual_param_shapes(width=WIDTH, layers=LAYERS, patch=PATCH, res=RES, out_dim=OUT_DIM, prefix="visual."):
    n_tok = (res // patch) ** 2 + 1
    S = {"class_embedding": (width,), "positional_embedding": (n_tok, width), "proj": (width, out_dim),
         "conv1.weight": (width, 3, patch, patch), "ln_pre.weight": (width,), "ln_pre.bias": (width,)}
    for i in range(layers):
        p = f"transformer.resblocks.{i}"
        S.update({f"{p}.attn.in_proj_weight": (3 * width, width), f"{p}.attn.in_proj_bias": (3 * width,),
                  f"{p}.attn.out_proj.weight": (width, width), f"{p}.attn.out_proj.bias": (width,),
                  f"{p}.ln_1.weight": (width,), f"{p}.ln_1.bias": (width,),
                  f"{p}.mlp.c_fc.weight": (4 * width, width), f"{p}.mlp.c_fc.bias": (4 * width,),
                  f"{p}.mlp.c_proj.weight": (width, 4 * width), f"{p}.mlp.c_proj.bias": (width,),
                  f"{p}.ln_2.weight": (width,), f"{p}.ln_2.bias": (width,)})
    S.update({"ln_post.weight": (width,), "ln_post.bias": (width,)})
    return {prefix + k: v for k, v in S.items()}


def attention(P, pre, x, heads):
    """nn.MultiheadAttention(E, heads)(x, x, x, need_weights=False)[0] for x [L, B, E] (sequence first, as CLIP calls it)."""
    L, B, E = x.shape
    qkv = F.linear(x, P[f"{pre}.in_proj_weight"], P[f"{pre}.in_proj_bias"])
    q, k, v = qkv.chunk(3, dim=-1)
    d = E // heads
    sh = lambda t: t.reshape(L, B * heads, d).transpose(0, 1)  # noqa: E731  [B*heads, L, d]
    q, k, v = sh(q) * (d ** -0.5), sh(k), sh(v)
    a = torch.softmax(q @ k.transpose(1, 2), dim=-1) @ v       # [B*heads, L, d]
    a = a.transpose(0, 1).reshape(L, B, E)
    return F.linear(a, P[f"{pre}.out_proj.weight"], P[f"{pre}.out_proj.bias"])


def encode_image(P, image, heads=HEADS, prefix="visual.", taps=None):
    """VisionTransformer.forward: image [B,3,R,R] (CLIP-normalised) -> [B, out_dim]."""
    g = lambda k: P[prefix + k]  # noqa: E731
    patch = g("conv1.weight").shape[-1]
    x = F.conv2d(image, g("conv1.weight"), stride=patch)                                 # [B, width, grid, grid]
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)                           # [B, grid^2, width]
    cls = g("class_embedding").to(x.dtype) + torch.zeros(x.shape[0], 1, x.shape[-1], dtype=x.dtype)
    x = torch.cat([cls, x], dim=1) + g("positional_embedding")
    width = x.shape[-1]
    x = F.layer_norm(x, (width,), g("ln_pre.weight"), g("ln_pre.bias"), 1e-5)
    x = x.permute(1, 0, 2)                                                               # [L, B, width]
    i = 0
    while f"{prefix}transformer.resblocks.{i}.ln_1.weight" in P:
        p = f"transformer.resblocks.{i}"
        h = F.layer_norm(x, (width,), g(f"{p}.ln_1.weight"), g(f"{p}.ln_1.bias"), 1e-5)
        x = x + attention(P, prefix + p + ".attn", h, heads)
        h = F.layer_norm(x, (width,), g(f"{p}.ln_2.weight"), g(f"{p}.ln_2.bias"), 1e-5)
        h = F.linear(h, g(f"{p}.mlp.c_fc.weight"), g(f"{p}.mlp.c_fc.bias"))
        h = h * torch.sigmoid(1.702 * h)
        x = x + F.linear(h, g(f"{p}.mlp.c_proj.weight"), g(f"{p}.mlp.c_proj.bias"))
        if taps is not None:
            taps[i] = x
        i += 1
    x = x.permute(1, 0, 2)
    x = F.layer_norm(x[:, 0, :], (width,), g("ln_post.weight"), g("ln_post.bias"), 1e-5)
    return x @ g("proj")
