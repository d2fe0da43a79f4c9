"""SEAN inpainting stage on the MI355X kernels - SURVEY.md section 8 row f4.

Host-side mirror of what models/Alignment.py:123-131 calls: `encode_sean` / `decode_sean`
(models/sean_codes/models/pix2pix_model.py:299-325) around `Pix2PixModel(SEAN_OPT).netG`, a `SPADEGenerator`
(networks/generator.py:14-110) of seven `SPADEResnetBlock`s (networks/architecture.py:21-97) whose normalisation is
`ACE` (networks/normalization.py:70-208: region-adaptive styles blended with `SPADE`, :211-257) and whose style
encoder is `Zencoder` (architecture.py:155-207).  Same module tree and state-dict keys as the reference
(`netG.Zencoder.model.1.weight`, `netG.head_0.conv_0.weight_orig` / `weight_u` / `weight_v` (torch's spectral_norm),
`netG.up_0.ace_s.fc_mu7.weight`, `netG.up_1.ace_0.Spade.mlp_gamma.bias`, ...), configuration SEAN_OPT
(pix2pix_model.py:328-339).  Inference only.

How it runs here (one batched pass for all decodes of a call, `group` consecutive decodes sharing a target mask - the
two images of a pair, Alignment.py:130-131):

* Every 3x3 convolution whose input is constant per segmentation label - `fc` on the one-hot map, SPADE's `mlp_shared`,
  ACE's `conv_gamma` / `conv_beta` on `middle_avg` - is nine table lookups per output (hf_label_conv3x3_f32, csrc/sean.hip);
  the tables of the style branch are one 19-column-per-sample GEMM per ACE (weights streamed once), the nineteen
  `fc_mu` layers of ALL fifteen styled ACEs one grouped launch.  The reference's 1.5 TFLOP of dense convolutions on
  `middle_avg` per decode are gone.
* SPADE's `mlp_gamma` / `mlp_beta` are one conv with 2C outputs, computed once per target mask; the dense convs
  (`conv_0`, `conv_1`, `conv_s`, the style encoder) run on the library's conv kernels (fp16 matrix cores where the shape
  allows) with spectral normalisation folded into the weights at load time, bias / residual / LeakyReLU in their epilogues.
* ACE's tail (noise, inference BatchNorm, blend, modulation, the block's LeakyReLU) is one pass (hf_ace_modulate_f32).
* Zencoder: reflection padding as torch glue, InstanceNorm + LeakyReLU on hf_layernorm_f32 (one row per plane), the
  transposed conv as a 3x3 conv of the zero-inserted input, tanh + per-region average pooling in one kernel.
"""
import torch
import torch.nn.functional as F
from torch import nn

from . import _marshal as M
from ._runtime import lib, require_gpu, stream
from .encoders._fused import FrozenPlanMixin, PreparedConv, conv, prep_conv

N_LABELS = 19  # SEAN_OPT.semantic_nc / label_nc


class SpectralConv(nn.Module):
    """nn.Conv2d under torch.nn.utils.spectral_norm (architecture.py:44-49): parameters `bias`, `weight_orig`, buffers
    `weight_u`, `weight_v`; in eval mode the weight is weight_orig / (u . W_mat v)."""

    def __init__(self, cin, cout, k, bias=True):
        super().__init__()
        self.k = k
        if bias:
            self.bias = nn.Parameter(torch.zeros(cout))
        else:
            self.register_parameter("bias", None)
        self.weight_orig = nn.Parameter(torch.empty(cout, cin, k, k).normal_(0, 0.02))
        self.register_buffer("weight_u", F.normalize(torch.randn(cout), dim=0))
        self.register_buffer("weight_v", F.normalize(torch.randn(cin * k * k), dim=0))

    def normalized_weight(self):
        w = self.weight_orig.detach()
        sigma = torch.dot(self.weight_u, torch.mv(w.reshape(w.shape[0], -1), self.weight_v))
        return w / sigma


class SPADE(nn.Module):  # normalization.py:211-257 (norm_G 'spadesyncbatch3x3')
    def __init__(self, norm_nc, label_nc=N_LABELS, hidden=128):
        super().__init__()
        self.param_free_norm = nn.BatchNorm2d(norm_nc, affine=False)  # constructed, never applied (:247-257)
        self.mlp_shared = nn.Sequential(nn.Conv2d(label_nc, hidden, 3, padding=1), nn.ReLU())
        self.mlp_gamma = nn.Conv2d(hidden, norm_nc, 3, padding=1)
        self.mlp_beta = nn.Conv2d(hidden, norm_nc, 3, padding=1)


class ACE(nn.Module):  # normalization.py:70-208
    def __init__(self, norm_nc, use_rgb=True, style=512, hidden=128):
        super().__init__()
        self.norm_nc, self.use_rgb, self.style, self.hidden = norm_nc, use_rgb, style, hidden
        STYLE_LEN = style
        self.Spade = SPADE(norm_nc, hidden=hidden)
        self.blending_gamma = nn.Parameter(torch.zeros(1))
        self.blending_beta = nn.Parameter(torch.zeros(1))
        self.noise_var = nn.Parameter(torch.zeros(norm_nc))
        self.param_free_norm = nn.BatchNorm2d(norm_nc, affine=False)  # SynchronizedBatchNorm2d: F.batch_norm in eval mode
        if use_rgb:
            for j in range(N_LABELS):
                setattr(self, f"fc_mu{j}", nn.Linear(STYLE_LEN, STYLE_LEN))
            self.conv_gamma = nn.Conv2d(STYLE_LEN, norm_nc, 3, padding=1)
            self.conv_beta = nn.Conv2d(STYLE_LEN, norm_nc, 3, padding=1)
        self._plan = None

    def plan(self):
        if self._plan is None:
            L, st = lib(), stream()
            bn, sp = self.param_free_norm, self.Spade
            ones, zeros = torch.ones_like(bn.running_var), torch.zeros_like(bn.running_var)
            scale, shift = M.bn_fold(L, st, ones, zeros, bn.running_mean, bn.running_var, bn.eps)
            c, N_HIDDEN, STYLE_LEN = self.norm_nc, self.hidden, self.style
            w_sh = sp.mlp_shared[0].weight.detach()  # [128, 19, 3, 3] -> table [(tap, c)][label]
            w_gb = torch.cat([sp.mlp_gamma.weight.detach(), sp.mlp_beta.weight.detach()], 0).contiguous()
            p = {"bn": (scale, shift), "noise_var": self.noise_var.detach(),
                 "blend": torch.cat([self.blending_gamma.detach(), self.blending_beta.detach()]).contiguous(),
                 "t_shared": w_sh.permute(2, 3, 0, 1).reshape(9 * N_HIDDEN, N_LABELS).contiguous(),
                 "b_shared": sp.mlp_shared[0].bias.detach(),
                 "w_gb": PreparedConv(M.conv_prepare(L, st, w_gb), 3),
                 "b_gb": torch.cat([sp.mlp_gamma.bias.detach(), sp.mlp_beta.bias.detach()]).contiguous()}
            if self.use_rgb:
                # the table GEMM: rows (tap, gamma|beta channel), K = 512 style features -> a 1x1 "conv" over 19*B pixels
                w = torch.cat([self.conv_gamma.weight.detach(), self.conv_beta.weight.detach()], 0)      # [2C, 512, 3, 3]
                w = w.permute(2, 3, 0, 1).reshape(9 * 2 * c, STYLE_LEN, 1, 1).contiguous()
                p["w_table"] = PreparedConv(M.conv_prepare(L, st, w), 1)
                p["b_avg"] = torch.cat([self.conv_gamma.bias.detach(), self.conv_beta.bias.detach()]).contiguous()
            self._plan = p
        return self._plan

    def fc_mu_weights(self):
        """([19, 512 (in), 512 (out)], [19, 512]): the nineteen fc_mu layers in grouped-1x1-conv layout."""
        ws = torch.stack([getattr(self, f"fc_mu{j}").weight.detach().t() for j in range(N_LABELS)])
        bs = torch.stack([getattr(self, f"fc_mu{j}").bias.detach() for j in range(N_LABELS)])
        return ws, bs

    def forward(self, x, labels, mu, noise, group, slope, x_up=False):
        """x [D,C,H,W]; labels int32 [D/group,H,W]; mu [1,512,D,19] (relu(fc_mu_j(code_j)) of every sample, GEMM layout) or
        None; noise [D,H,W] | None.  Returns LeakyReLU_slope(ACE(x)) (slope 1: the shortcut branch, no activation).
        x_up: x is the HALF-resolution tensor [D,C,H/2,W/2]; the generator's nearest x2 up-sampling is read in place by the
        tail kernel (both tail kernels take it)."""
        L, st = lib(), stream()
        p = self.plan()
        d, c = x.shape[:2]
        h, w = labels.shape[-2:]
        N_HIDDEN = self.hidden
        actv = M.label_conv3x3(L, st, labels, p["t_shared"], p["b_shared"], N_HIDDEN, relu=True)      # [D/group,128,H,W]
        sp = conv(actv, p["w_gb"], 3, 1, bias=p["b_gb"])                                              # [D/group,2C,H,W]
        avg = None
        if self.use_rgb:
            table = conv(mu, p["w_table"], 1, 1)                                                      # [1, 9*2C, D, 19]
            if M.ace_modulate_table_supported(h, w):  # the avg planes are looked up inside the tail kernel, never stored
                return M.ace_modulate_table(L, st, x, noise, p["noise_var"], p["bn"][0], p["bn"][1], labels,
                                            table.reshape(9 * 2 * c, d * N_LABELS), p["b_avg"], sp, p["blend"], group=group, slope=slope,
                                            x_up=x_up)
            avg = M.label_conv3x3(L, st, labels, table.reshape(9 * 2 * c, d * N_LABELS), p["b_avg"], 2 * c, batch=d,
                                  cols_per_sample=N_LABELS, group=group)
        return M.ace_modulate(L, st, x, noise, p["noise_var"], p["bn"][0], p["bn"][1], avg, sp, p["blend"] if avg is not None else None,
                              group=group, slope=slope, x_up=x_up)


class SPADEResnetBlock(nn.Module):  # architecture.py:21-97
    def __init__(self, fin, fout, use_rgb=True, style=512, hidden=128):
        super().__init__()
        self.fin, self.fout, self.use_rgb = fin, fout, use_rgb
        self.learned_shortcut = fin != fout
        fmid = min(fin, fout)
        self.conv_0 = SpectralConv(fin, fmid, 3)
        self.conv_1 = SpectralConv(fmid, fout, 3)
        if self.learned_shortcut:
            self.conv_s = SpectralConv(fin, fout, 1, bias=False)
        self.ace_0 = ACE(fin, use_rgb, style, hidden)
        self.ace_1 = ACE(fmid, use_rgb, style, hidden)
        if self.learned_shortcut:
            self.ace_s = ACE(fin, use_rgb, style, hidden)
        self._plan = None

    def aces(self):  # in call order (:67-93)
        return ([self.ace_s] if self.learned_shortcut else []) + [self.ace_0, self.ace_1]

    def forward(self, x, labels, mus, noises, group, final_lrelu=False, x_up=False):
        """mus / noises: per ACE of `aces()`.  final_lrelu: the LeakyReLU(0.2) SPADEGenerator applies to the LAST block's
        output before conv_img (generator.py:108) rides in conv_1's epilogue.  x_up (learned-shortcut blocks only: x is read
        by ace_s and ace_0 and by nothing else): x arrives at HALF resolution and the generator's `self.up` happens inside the
        two ACE tails - the up-sampled tensor (4x the bytes) is neither written nor re-read."""
        if x_up and not self.learned_shortcut:
            raise ValueError("x_up needs a learned shortcut (the identity shortcut adds x itself)")
        L, st = lib(), stream()
        if self._plan is None:
            prep = lambda m: PreparedConv(M.conv_prepare(L, st, m.normalized_weight().contiguous()), m.k)  # noqa: E731
            self._plan = {"w0": prep(self.conv_0), "w1": prep(self.conv_1), "ws": prep(self.conv_s) if self.learned_shortcut else None}
        p = self._plan
        k = 0
        x_s = x
        if self.learned_shortcut:
            x_s = conv(self.ace_s(x, labels, mus[0], noises[0], group, 1.0, x_up=x_up), p["ws"], 1, 1)
            k = 1
        dx = self.ace_0(x, labels, mus[k], noises[k], group, 0.2, x_up=x_up)
        dx = conv(dx, p["w0"], 3, 1, bias=self.conv_0.bias.detach())
        dx = self.ace_1(dx, labels, mus[k + 1], noises[k + 1], group, 0.2)
        if final_lrelu:
            return conv(dx, p["w1"], 3, 1, bias=self.conv_1.bias.detach(), residual=x_s, act=M.ACT_LRELU | M.ACT_RESIDUAL_FIRST, alpha=0.2)
        return conv(dx, p["w1"], 3, 1, bias=self.conv_1.bias.detach(), residual=x_s)


class Zencoder(nn.Module):  # architecture.py:155-207 (input_nc 3, output_nc 512, ngf 32, n_downsampling 2)
    def __init__(self, style=512, widths=(32, 64, 128, 256)):
        super().__init__()
        lr = lambda: nn.LeakyReLU(0.2, False)  # noqa: E731
        z0, z1, z2, z3 = widths
        self.model = nn.Sequential(
            nn.ReflectionPad2d(1), nn.Conv2d(3, z0, 3, padding=0), nn.InstanceNorm2d(z0), lr(),
            nn.Conv2d(z0, z1, 3, stride=2, padding=1), nn.InstanceNorm2d(z1), lr(),
            nn.Conv2d(z1, z2, 3, stride=2, padding=1), nn.InstanceNorm2d(z2), lr(),
            nn.ConvTranspose2d(z2, z3, 3, stride=2, padding=1, output_padding=1), nn.InstanceNorm2d(z1), lr(),
            nn.ReflectionPad2d(1), nn.Conv2d(z3, style, 3, padding=0), nn.Tanh())
        self._plan = None

    def forward(self, image, labels):
        """image [N,3,H,W]; labels int32 [N,H/2,W/2] (the label map at the codes' resolution) -> [N,19,512]."""
        L, st = lib(), stream()
        m = self.model
        if self._plan is None:
            prep = lambda w: PreparedConv(M.conv_prepare(L, st, w.contiguous()), 3)  # noqa: E731
            # ConvTranspose2d(k 3, s 2, p 1, op 1) = 3x3 / pad 1 conv of the zero-inserted input [.., 2H, 2W] (x at even
            # positions) with the kernel transposed in (cin, cout) and flipped
            wt = m[10].weight.detach().permute(1, 0, 2, 3).flip(2, 3)
            self._plan = {"w1": prep(m[1].weight.detach()), "w4": prep(m[4].weight.detach()), "w7": prep(m[7].weight.detach()),
                          "w10": prep(wt), "w14": prep(m[14].weight.detach())}
        p = self._plan
        inorm = lambda t: M.layernorm(L, st, t, t.shape[-2] * t.shape[-1], eps=1e-5, lrelu=True, alpha=0.2)  # noqa: E731
        x = conv(F.pad(image, (1, 1, 1, 1), mode="reflect"), p["w1"], 3, 1, bias=m[1].bias.detach())
        x = inorm(x[:, :, 1:-1, 1:-1].contiguous())
        x = inorm(conv(x, p["w4"], 3, 2, bias=m[4].bias.detach()))
        x = inorm(conv(x, p["w7"], 3, 2, bias=m[7].bias.detach()))
        n, c, h, w = x.shape
        z = x.new_zeros((n, c, 2 * h, 2 * w))
        z[:, :, ::2, ::2] = x
        x = inorm(conv(z, p["w10"], 3, 1, bias=m[10].bias.detach()))
        x = conv(F.pad(x, (1, 1, 1, 1), mode="reflect"), p["w14"], 3, 1, bias=m[14].bias.detach())   # valid part: the interior
        return M.region_mean(L, st, x, labels, crop=1, act_tanh=True)  # [N, 19, style]


class SPADEGenerator(FrozenPlanMixin, nn.Module):  # generator.py:14-110, num_upsampling_layers 'normal'
    UP_BEFORE = (1, 3, 4, 5, 6)  # nn.Upsample(scale_factor=2) in front of these blocks (:85-104)

    def __init__(self, ngf=64, style=512, hidden=128, size=256, zencoder_widths=(32, 64, 128, 256)):
        """Defaults = SEAN_OPT (ngf 64, crop_size 256) with the reference's hard-coded style_length 512 / nhidden 128;
        smaller values build the scaled-down model of the CPU interpreter tests."""
        super().__init__()
        self.ngf, self.style, self.size = ngf, style, size
        n = ngf
        self.BLOCKS = [("head_0", 16 * n, 16 * n, True), ("G_middle_0", 16 * n, 16 * n, True), ("G_middle_1", 16 * n, 16 * n, True),
                       ("up_0", 16 * n, 8 * n, True), ("up_1", 8 * n, 4 * n, True), ("up_2", 4 * n, 2 * n, True),
                       ("up_3", 2 * n, 1 * n, False)]
        self.Zencoder = Zencoder(style, zencoder_widths)
        self.fc = nn.Conv2d(N_LABELS, 16 * ngf, 3, padding=1)
        for name, fin, fout, rgb in self.BLOCKS:
            setattr(self, name, SPADEResnetBlock(fin, fout, rgb, style, hidden))
        self.conv_img = nn.Conv2d(ngf, 3, 3, padding=1)
        self._plan = None
        # tests: callable(d, sizes) -> list of the 18 ACE noise tensors [d, r, r] instead of fresh torch.randn draws
        self.noise_source = None

    def blocks(self):
        return [getattr(self, name) for name, *_ in self.BLOCKS]

    def invalidate(self):
        super().invalidate()
        self._plan = None

    def plan(self):
        if self._plan is None:
            L, st = lib(), stream()
            styled = [a for blk in self.blocks() for a in blk.aces() if a.use_rgb]
            ws, bs = zip(*[a.fc_mu_weights() for a in styled])
            self._plan = {
                "t_fc": self.fc.weight.detach().permute(2, 3, 0, 1).reshape(9 * 16 * self.ngf, N_LABELS).contiguous(),
                "mu_w": PreparedConv(torch.cat(ws, 0).unsqueeze(1).contiguous(), 1),  # [15*19, 1, 512, 512] grouped 1x1 conv weights
                "mu_b": torch.cat(bs, 0).contiguous(),                # [15*19, 512]
                "n_styled": len(styled),
                "w_img": prep_conv(self.conv_img, pad=True)}  # 64 -> 3: zero filters up to 64, sliced off again
        return self._plan

    def decode(self, codes, target_labels, group=1, noise=None, taps=None):
        """codes [D,19,512]: per-label style codes of every image to render (decode_sean's obj_dic); target_labels long or
        int [D/group,1,256,256]; noise: None (fresh draws, as the reference) or the list of 18 ACE noise tensors [D,H,W] in
        call order.  Returns [D,3,256,256] in (-1, 1)."""
        require_gpu(codes, target_labels)
        L, st = lib(), stream()
        p = self.plan()
        d = codes.shape[0]
        S, NGF, STYLE_LEN = self.size, self.ngf, self.style
        if target_labels.shape[0] * group != d or tuple(target_labels.shape[1:]) != (1, S, S):
            raise ValueError(f"target_labels must be [{d}/{group},1,{S},{S}]; got {tuple(target_labels.shape)}")
        lab = target_labels[:, 0].to(torch.int32)
        labels = {S >> k: lab[:, ::1 << k, ::1 << k].contiguous() for k in range(6)}  # F.interpolate(seg, mode='nearest')
        # every fc_mu of every styled ACE in one grouped launch: mu[a, j, b] = relu(W_aj code[b, j] + b_aj)
        n_st = p["n_styled"]
        xg = codes.permute(1, 0, 2).reshape(1, N_LABELS, d, STYLE_LEN).expand(n_st, -1, -1, -1)
        xg = xg.reshape(n_st * N_LABELS, d, STYLE_LEN, 1, 1).contiguous()
        mu = conv(xg, p["mu_w"], 1, 1, bias=p["mu_b"], act=M.ACT_LRELU, alpha=0.0, groups=n_st * N_LABELS, x_shared=False)
        mu = mu.reshape(n_st, N_LABELS, d, STYLE_LEN).permute(0, 3, 2, 1).contiguous()            # [15, 512, D, 19]
        aces = [a for blk in self.blocks() for a in blk.aces()]
        if noise is None and self.noise_source is not None:
            res_, sizes_ = S // 32, []
            for i, blk in enumerate(self.blocks()):
                res_ = res_ * 2 if i in self.UP_BEFORE else res_
                sizes_ += [res_] * len(blk.aces())
            noise = self.noise_source(d, sizes_)
        if noise is None:
            sizes = []
            res = S // 32
            for i, blk in enumerate(self.blocks()):
                if i in self.UP_BEFORE:
                    res *= 2
                sizes += [res] * len(blk.aces())
            flat = torch.randn(sum(d * r * r for r in sizes), device=codes.device)
            noise, o = [], 0
            for r in sizes:
                noise.append(flat[o:o + d * r * r].view(d, r, r))
                o += d * r * r
        elif len(noise) != len(aces):
            raise ValueError(f"noise must list {len(aces)} tensors")
        x = M.label_conv3x3(L, st, labels[S // 32], p["t_fc"], self.fc.bias.detach(), 16 * NGF)     # [D/group,1024,8,8]
        if group > 1:
            x = x.repeat_interleave(group, 0)
        res, ai, si = S // 32, 0, 0
        for i, blk in enumerate(self.blocks()):
            x_up = False
            if i in self.UP_BEFORE:
                res *= 2
                x_up = blk.learned_shortcut and res % 4 == 0   # the block's ACE tails read the low-resolution x in place
                if not x_up:
                    x = M.upsample_nearest(L, st, x, res, res)
            n_a = len(blk.aces())
            mus = []
            for a in blk.aces():
                mus.append(mu[si:si + 1] if a.use_rgb else None)
                si += 1 if a.use_rgb else 0
            x = blk(x, labels[res], mus, noise[ai:ai + n_a], group, final_lrelu=(i == len(self.BLOCKS) - 1), x_up=x_up)
            ai += n_a
            if taps is not None:
                taps[self.BLOCKS[i][0]] = x
        x = conv(x, p["w_img"], 3, 1, bias=self.conv_img.bias.detach())
        return M.tanh(L, st, x)


class SeanModel(nn.Module):
    """Pix2PixModel(SEAN_OPT) as HairFast uses it (Alignment.py:29-30, 126-131): `netG` + the per-label median style
    codes of load_average_feature (pix2pix_model.py:268-293; the nineteen
    models/sean_codes/styles_test/mean_style_code/median/<label>/ACE.npy files as one [19,512] tensor `mean_codes`)."""

    def __init__(self, mean_codes=None, **generator_sizes):
        """mean_codes None builds the module without them (state-dict layout checks, encode-only use): `decode` then
        raises - the reference ALWAYS starts from load_average_feature() and a label of the target mask that the source
        image lacks is common, so an implicit all-zero default would render it from fc_mu(0) without any error."""
        super().__init__()
        self.netG = SPADEGenerator(**generator_sizes)
        if mean_codes is not None:
            mean_codes = torch.as_tensor(mean_codes).float()
            if tuple(mean_codes.shape) != (N_LABELS, self.netG.style):
                raise ValueError(f"sean_mean_codes: expected [{N_LABELS}, {self.netG.style}] (one median ACE.npy code per label, "
                                 f"pix2pix_model.py:268-293), got {tuple(mean_codes.shape)}")
            mean_codes = mean_codes.clone()
        self.register_buffer("mean_codes", mean_codes, persistent=False)

    @torch.inference_mode()
    def encode(self, images, labels):
        """encode_sean (:299-307): images [N,3,256,256], labels long [N,1,256,256] -> style codes [N,19,512]."""
        require_gpu(images, labels)
        lab = labels[:, 0, ::2, ::2].to(torch.int32).contiguous()  # F.interpolate(segmap, 128, 'nearest') of the one-hot map
        return self.netG.Zencoder(images.float(), lab)

    @torch.inference_mode()
    def decode(self, image_code, target_mask, group=1, noise=None, taps=None):
        """decode_sean (:310-325) for D codes at once: a label's code is the image's where it is not all zero (the label
        occurs in the image), the median code otherwise."""
        if self.mean_codes is None:
            raise ValueError("SeanModel.decode needs the per-label median style codes (sean_mean_codes [19,512]: "
                             "models/sean_codes/styles_test/mean_style_code/median/<label>/ACE.npy, pix2pix_model.py:268-293)")
        absent = (image_code == 0).all(dim=-1, keepdim=True)
        codes = torch.where(absent, self.mean_codes.to(image_code.device).unsqueeze(0).expand_as(image_code), image_code)
        return self.netG.decode(codes.contiguous(), target_mask, group=group, noise=noise, taps=taps)

    @torch.inference_mode()
    def inpaint_pairs(self, images_256, labels, target_masks, noise=None):
        """The SEAN step of align_images (Alignment.py:123-131) for P pairs: images_256 [2P,3,256,256] in [0,1] and their
        label maps [2P,1,256,256] (pair p = rows 2p, 2p+1), target_masks [P,1,256,256] -> [2P,3,256,256] in (-1,1): both
        images of every pair re-rendered on the pair's target mask."""
        codes = self.encode(images_256, labels)
        return self.decode(codes, target_masks, group=2, noise=noise)
