"""PostProcessModel on the MI355X kernels - host-side mirror of
models/Encoders.py:106-137 (`PostProcessModel`), :13-32 (`ModulationModule`), :35-57
(`FeatureiResnet`) and models/Net.py:396-477 (`FeatureEncoderMult(fs_layers=[9])`) - and, built from the same
ModulationModule, the two latent-space models of row f4: `RotateModel` (:60-72, complete) and
`ClipBlendingModel` (:75-103: its own parameters; the CLIP ViT-B/32 image tower it embeds the two images with - an
un-vendored dependency of the reference, `clip @ git+...`, requirements.txt:6 - runs natively too since round 3:
hairfastgan_amd/clip_vit.py, handed in as `clip_image_embed`).

It sits immediately before the last generator call of a swap (models/Blending.py:66-68):
`S_final, F_final = post_process(I_1, I_blend_256)`, then `generator([S_final], start_layer=5,
end_layer=8, layer_in=F_final)`.  774 GFLOP per triple - more than the generator's share of a swap -
almost all of it 3x3 convolutions at 64x64 with 512...1024 channels, which run on the fp16
matrix-core kernel of csrc/convh_enc.hip (IBasicBlock of encoders/fs_encoder.py: BN before the conv
as an affine on real pixels, BN + PReLU / residual in the epilogue).

Same parameter names as the reference (`encoder_face.*`, `to_feature.res_blocks.*`,
`to_latent_1.*`, `to_latent_2.*`; `latent_avg` is a tensor attribute there, loaded from
pretrained_models/PostProcess/latent_avg.pt, and a non-persistent buffer here).

One scheduling change: the reference encodes source and target with two batch-1 calls of the same
`encoder_face` (:120-121); here they are one batch-2 forward (frozen weights, independent samples).
Inference only.
"""
import torch
from torch import nn

from .. import _marshal as M
from .._runtime import lib, require_gpu, stream
from ._fused import FrozenPlanMixin, conv, conv_pair, fold_bn, prep_conv
from .fs_encoder import _IRESNET50, IBasicBlock, _make_layer, run_block_chain


class FeatureEncoderMult(FrozenPlanMixin, nn.Module):  # models/Net.py:396-477 with fs_layers=[9], ranks=None
    def __init__(self, n_styles=18):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(3, 64, 3, 1, 1, bias=False), nn.BatchNorm2d(64, eps=1e-05), nn.PReLU(64))
        inpl = 64
        for li, (planes, blocks) in enumerate(_IRESNET50):
            setattr(self, f"block_{li + 1}", _make_layer(inpl, planes, blocks))
            inpl = planes
        # max(fs_layers) > 7: scale 2 -> the content branches off block_2 (128 channels, 64x64), 3x3 stride 1
        self.content_layer = nn.ModuleList([nn.Sequential(
            nn.BatchNorm2d(128, eps=1e-05), nn.Conv2d(128, 512, 3, 1, 1, bias=False), nn.BatchNorm2d(512, eps=1e-05),
            nn.PReLU(num_parameters=512), nn.Conv2d(512, 512, 3, 1, 1, bias=False), nn.BatchNorm2d(512, eps=1e-05))])
        self.avg_pool = nn.AdaptiveAvgPool2d((3, 3))
        self.styles = nn.ModuleList([nn.Linear(960 * 9, 512) for _ in range(n_styles)])
        self._plan = None

    def forward(self, x):
        """x [B,3,256,256] -> (latents [B,18,512], [content [B,512,64,64]])."""
        require_gpu(x)
        if tuple(x.shape[-2:]) != (256, 256):
            raise NotImplementedError("transform_to_256 (torchvision Resize, models/Net.py:12-14) is the identity only on 256x256 "
                                      "inputs - which is what Blending.py:66 passes")
        L, st = lib(), stream()
        if self._plan is None:
            c, cl = self.conv, self.content_layer[0]
            self._plan = {
                "w_in": prep_conv(c[0], pad=True), "bn_in": fold_bn(c[1]), "slope_in": c[2].weight.detach(),
                "c_bn0": fold_bn(cl[0]), "c_w1": prep_conv(cl[1]), "c_bn2": fold_bn(cl[2]),
                "c_slope": cl[3].weight.detach(), "c_w4": prep_conv(cl[4]), "c_bn5": fold_bn(cl[5]),
                "head_w": torch.cat([m.weight.detach() for m in self.styles], 0).contiguous(),
                "head_b": torch.cat([m.bias.detach() for m in self.styles], 0).contiguous()}
        p = self._plan
        x = conv(x, p["w_in"], 3, 1, out_scale=p["bn_in"][0], bias=p["bn_in"][1], act=M.ACT_PRELU, slope=p["slope_in"])
        b = x.shape[0]
        pooled = x.new_empty((b, 960, 3, 3))
        c_off, content = 0, None
        xs = None
        for li in range(4):
            nxt_layer = getattr(self, f"block_{li + 2}") if li < 3 else None
            x, xs = run_block_chain(getattr(self, f"block_{li + 1}"), x, xs, after=None if nxt_layer is None else nxt_layer[0])
            if li == 1:
                content = conv_pair(x, p["c_w1"], dict(in_scale=p["c_bn0"][0], in_shift=p["c_bn0"][1], out_scale=p["c_bn2"][0],
                                                       bias=p["c_bn2"][1], act=M.ACT_PRELU, slope=p["c_slope"]),
                                    p["c_w4"], 1, dict(out_scale=p["c_bn5"][0], bias=p["c_bn5"][1]))
            M.adaptive_avgpool_into(L, st, pooled, x, c_off)
            c_off += x.shape[1]
        out = M.linear(L, st, pooled.reshape(b, -1), p["head_w"], p["head_b"], 1.0)
        return out.reshape(b, len(self.styles), 512), [content]


_STACK_GENERATION = [0]  # bumped by every ModulationModule.load_state_dict (see modulation_stack)


def _bump_stack_generation(_module, _incompatible_keys):  # a module-level function: a lambda hook would make the module unpicklable
    _STACK_GENERATION[0] += 1


def _param_version(p):
    """What tells a stacked copy of `p` is stale: the storage, the dtype and the in-place update counter (inference tensors
    have none: they cannot be updated in place).  Writes through `param.data` bypass autograd's counters by design and
    stay invisible here as anywhere else in torch: bump `_STACK_GENERATION[0]` after such a write."""
    try:
        ver = p._version
    except RuntimeError:
        ver = None
    return (p.data_ptr(), p.dtype, ver)


class ModulationModule(nn.Module):  # models/Encoders.py:13-32
    def __init__(self, layernum, last=False, inp=512, middle=512):
        super().__init__()
        # modulation_stack caches stacked copies of the branch weights on the stack's first module: a load into ANY
        # ModulationModule outdates every cached stack (the generation counter below)
        self.register_load_state_dict_post_hook(_bump_stack_generation)
        self.layernum, self.last = layernum, last
        self.fc = nn.Linear(512, 512)
        self.norm = nn.LayerNorm([layernum, 512], elementwise_affine=False)
        self.gamma_function = nn.Sequential(nn.Linear(inp, middle), nn.LayerNorm([middle]), nn.LeakyReLU(), nn.Linear(middle, 512))
        self.beta_function = nn.Sequential(nn.Linear(inp, middle), nn.LayerNorm([middle]), nn.LeakyReLU(), nn.Linear(middle, 512))
        self.leakyrelu = nn.LeakyReLU()

    def forward(self, x, embedding):
        """x, embedding [B, layernum, 512] -> [B, layernum, 512]: 9 launches (3 weight-streaming linears per
        branch, LayerNorm + LeakyReLU fused, one modulation pass)."""
        require_gpu(x, embedding)
        L, st = lib(), stream()
        lin = lambda t, m: M.linear(L, st, t.reshape(-1, t.shape[-1]), m.weight.detach(), m.bias.detach(), 1.0)  # noqa: E731
        shape = x.shape
        h = M.layernorm(L, st, lin(x, self.fc), self.layernum * 512, eps=self.norm.eps)

        def mlp(f):
            mid = M.layernorm(L, st, lin(embedding, f[0]), f[1].normalized_shape[0], f[1].weight.detach(), f[1].bias.detach(),
                              eps=f[1].eps, lrelu=True, alpha=f[2].negative_slope)
            return lin(mid, f[3])

        out = M.modulate(L, st, h, mlp(self.gamma_function), mlp(self.beta_function), lrelu=not self.last,
                         alpha=self.leakyrelu.negative_slope)
        return out.reshape(shape)


def modulation_stack(mods, x, embedding):
    """`for mod in mods: x = mod(x, embedding)` (models/Encoders.py:67-69, 99-101, 123-127) with the gamma / beta branches of
    ALL modules computed up front: they only depend on `embedding`, which is the same for every module of a stack - ONE
    stacked first Linear ([2M * middle, inp] rows: module 0 gamma, module 0 beta, module 1 gamma, ...), ONE grouped
    LayerNorm + LeakyReLU over its columns (hf_layernorm_grouped_f32), then the 2M second Linears on column slices; the
    dependent chain per module is fc -> LayerNorm -> modulate.  27 instead of 45 launches for five modules; per output the
    same dot products in the same order: bit-identical to the module-by-module form."""
    require_gpu(x, embedding)
    L, st = lib(), stream()
    mods = list(mods)
    head = mods[0]
    plan = head.__dict__.get("_stack_plan")
    branches = [f for m in mods for f in (m.gamma_function, m.beta_function)]
    # the stacked copies follow their sources: module identities, load_state_dict, and - per stacked parameter - storage,
    # dtype and in-place version (param.data.copy_, .half(), an optimizer step, a sub-module's _load_from_state_dict)
    key = (tuple(id(m) for m in mods), _STACK_GENERATION[0], mods[0].fc.weight.device,
           tuple(_param_version(p) for f in branches for p in (f[0].weight, f[0].bias, f[1].weight, f[1].bias)))
    if plan is None or plan["key"] != key:
        plan = {"key": key,
                "w1": torch.cat([f[0].weight.detach() for f in branches], 0).contiguous(),
                "b1": torch.cat([f[0].bias.detach() for f in branches], 0).contiguous(),
                "ln_w": torch.stack([f[1].weight.detach() for f in branches]).contiguous(),
                "ln_b": torch.stack([f[1].bias.detach() for f in branches]).contiguous(),
                "middle": branches[0][1].normalized_shape[0], "eps": branches[0][1].eps, "slope": branches[0][2].negative_slope}
        head.__dict__["_stack_plan"] = plan
    G, middle = 2 * len(mods), plan["middle"]
    rows = embedding.reshape(-1, embedding.shape[-1])
    mid = M.linear(L, st, rows, plan["w1"], plan["b1"], 1.0)                                   # [R, G * middle]
    mid = M.layernorm(L, st, mid, middle, plan["ln_w"], plan["ln_b"], eps=plan["eps"], lrelu=True, alpha=plan["slope"], groups=G)
    shape = x.shape
    for i, m in enumerate(mods):
        gamma = M.linear(L, st, mid[:, (2 * i) * middle:(2 * i + 1) * middle], m.gamma_function[3].weight.detach(),
                         m.gamma_function[3].bias.detach(), 1.0)
        beta = M.linear(L, st, mid[:, (2 * i + 1) * middle:(2 * i + 2) * middle], m.beta_function[3].weight.detach(),
                        m.beta_function[3].bias.detach(), 1.0)
        h = M.linear(L, st, x.reshape(-1, x.shape[-1]), m.fc.weight.detach(), m.fc.bias.detach(), 1.0)
        h = M.layernorm(L, st, h, m.layernum * 512, eps=m.norm.eps)
        x = M.modulate(L, st, h, gamma, beta, lrelu=not m.last, alpha=m.leakyrelu.negative_slope).reshape(shape)
    return x


class FeatureiResnet(nn.Module):  # models/Encoders.py:35-57
    def __init__(self, blocks, inplanes=1024):
        super().__init__()
        res_blocks = {}
        for n, (planes, num_blocks) in enumerate(blocks, start=1):
            for k in range(1, num_blocks + 1):
                downsample = None
                if inplanes != planes:
                    downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, 1, bias=False), nn.BatchNorm2d(planes, eps=1e-05))
                res_blocks[f"res_block_{n}_{k}"] = IBasicBlock(inplanes, planes, 1, downsample)
                inplanes = planes
        self.res_blocks = nn.ModuleDict(res_blocks)

    def forward(self, x):
        return run_block_chain(self.res_blocks.values(), x)[0]  # block -> block hand-over of the pre-split first-conv input


class PostProcessModel(nn.Module):  # models/Encoders.py:106-137
    def __init__(self, latent_avg=None):
        super().__init__()
        self.encoder_face = FeatureEncoderMult()
        self.register_buffer("latent_avg", torch.zeros(18, 512) if latent_avg is None else latent_avg.clone(), persistent=False)
        self.to_feature = FeatureiResnet([[1024, 2], [768, 2], [512, 2]])
        self.to_latent_1 = nn.ModuleList([ModulationModule(18, i == 4) for i in range(5)])
        self.to_latent_2 = nn.ModuleList([ModulationModule(18, i == 4) for i in range(5)])

    @torch.inference_mode()
    def forward(self, source, target):
        """(source, target) [B,3,256,256] normalised -> (S_final [B,18,512], F_final [B,512,64,64])."""
        require_gpu(source, target)
        L, st = lib(), stream()
        b = source.shape[0]
        s_both, (f_both,) = self.encoder_face(torch.cat((source, target), 0))  # one batch-2B forward (the reference: two calls)
        s_face, s_hair = s_both[:b].contiguous(), s_both[b:].contiguous()
        dt_face = M.pixel_norm_dim1(L, st, s_face)
        dt_hair = M.pixel_norm_dim1(L, st, s_hair)
        dt_face = modulation_stack(self.to_latent_1, dt_face, s_hair)
        dt_hair = modulation_stack(self.to_latent_2, dt_hair, s_face)
        total = M.axpby(L, st, dt_face, 1.0, dt_hair, 1.0)                          # dt_face + dt_hair
        final_s = M.axpby(L, st, total, 0.1, self.latent_avg.reshape(-1), 1.0)      # latent_avg + 0.1 * (...), :131
        cat_f = torch.cat((f_both[:b], f_both[b:]), dim=1)                # [B,1024,64,64]
        final_f = self.to_feature(cat_f)
        return final_s, final_f


class RotateModel(nn.Module):  # models/Encoders.py:60-72
    """W+ rows 0..5 of the shape image rotated towards the pose of the face image (Alignment.py:61):
    latent_from + 0.1 * five ModulationModules(PixelNorm(latent_from) | latent_to).  15 linear launches + 11 small ones."""

    def __init__(self):
        super().__init__()
        self.modulation_module_list = nn.ModuleList([ModulationModule(6, i == 4) for i in range(5)])

    @torch.inference_mode()
    def forward(self, latent_from, latent_to):
        require_gpu(latent_from, latent_to)
        L, st = lib(), stream()
        latent_from, latent_to = latent_from.contiguous(), latent_to.contiguous()
        dt = M.pixel_norm_dim1(L, st, latent_from)
        dt = modulation_stack(self.modulation_module_list, dt, latent_to)
        return M.axpby(L, st, dt, 0.1, latent_from.reshape(-1), 1.0).reshape(latent_from.shape)  # latent_from + 0.1 * dt


class ClipBlendingModel(nn.Module):  # models/Encoders.py:75-103
    """S rows 6..17 of the blended image (Blending.py:60): latent_face + 0.1 * five ModulationModules(12, inp = 512 * 3,
    middle = 1024) of PixelNorm(latent_face) | [latent_color, CLIP(target_face), CLIP(hair_color)].

    image_embed: callable [B,3,224,224] (CLIP-normalised) -> [B,512], the reference's `clip_model.encode_image`
    (clip.load("ViT-B/32")): NOT part of this backend (an external package the reference pip-installs from git);
    everything else - pooling to 224^2, CLIP normalisation, the modulation stack - is.  State-dict keys as in the
    reference (`modulation_module_list.*`; its frozen `clip_model.*` entries belong to the injected callable)."""

    CLIP_MEAN, CLIP_STD = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)

    def __init__(self, image_embed=None):
        super().__init__()
        self.image_embed = image_embed
        self.modulation_module_list = nn.ModuleList([ModulationModule(12, i == 4, inp=512 * 3, middle=1024) for i in range(5)])

    def get_image_embed(self, image_tensor):  # :91-94
        if self.image_embed is None:
            raise NotImplementedError("ClipBlendingModel needs image_embed = the CLIP ViT-B/32 image encoder "
                                      "(clip_model.encode_image of the reference's un-vendored `clip` package)")
        x = torch.nn.functional.adaptive_avg_pool2d(image_tensor, (224, 224)) * 0.5 + 0.5
        key = (x.device, x.dtype)
        if self.__dict__.get("_norm_key") != key:  # created once (no host-to-device copy per call / inside a hipGraph capture)
            self.__dict__["_norm"] = (torch.tensor(self.CLIP_MEAN, device=x.device, dtype=x.dtype).view(1, 3, 1, 1),
                                      torch.tensor(self.CLIP_STD, device=x.device, dtype=x.dtype).view(1, 3, 1, 1))
            self.__dict__["_norm_key"] = key
        mean, std = self.__dict__["_norm"]
        return self.image_embed((x - mean) / std)

    @torch.inference_mode()
    def forward(self, latent_face, latent_color, target_face, hair_color):
        require_gpu(latent_face, latent_color)
        L, st = lib(), stream()
        # both images of every triple through the tower in ONE call (rows are independent: clip/model.py has no
        # cross-sample operation; the reference calls encode_image twice, Encoders.py:97-98)
        n = target_face.shape[0]
        embed = self.get_image_embed(torch.cat([target_face, hair_color], 0)).float()
        embed_face = embed[:n].unsqueeze(1).expand(-1, 12, -1)
        embed_color = embed[n:].unsqueeze(1).expand(-1, 12, -1)
        latent_in = torch.cat((latent_color, embed_face, embed_color), dim=-1).contiguous()
        latent_face = latent_face.contiguous()
        dt = M.pixel_norm_dim1(L, st, latent_face)
        dt = modulation_stack(self.modulation_module_list, dt, latent_in)
        return M.axpby(L, st, dt, 0.1, latent_face.reshape(-1), 1.0).reshape(latent_face.shape)
