"""Call surface of the reference's hair_swap.py for the part this backend covers.

`get_parser()` reproduces the reference's CLI defaults (hair_swap.py:108-133) so that
`Net(opts)` and the encoders are configured identically.  `HairFastHotPath` bundles the
three networks of the hot path (generator, e4e, FeatureStyle encoder) and replays the
hot-path CALL SCHEDULE of one `HairFast.swap`:
the same generator / encoder invocations, batch sizes and layer ranges that
Embedding -> Alignment -> Blending issue (models/Embedding.py:51-53,71-90,
models/Alignment.py:63,128-131, models/Blending.py:62,68).  The stages in between
(BiSeNet, SEAN, shape adaptor, CLIP blending, PostProcess) are out of scope of this
backend, so their outputs are replaced by tensors of the right shape; with the reference
installed they come from the reference's own modules bound to this backend
(INTEGRATION.md).
"""
import argparse
from pathlib import Path

import torch

from .encoders import Encoder4Editing, FSEncoder, get_latents
from .net import Net


def get_parser():
    parser = argparse.ArgumentParser(description="HairFast")
    parser.add_argument("--save_all_dir", type=Path, default=Path("output"))
    parser.add_argument("--size", type=int, default=1024)
    parser.add_argument("--ckpt", type=str, default="pretrained_models/StyleGAN/ffhq.pt")
    parser.add_argument("--channel_multiplier", type=int, default=2)
    parser.add_argument("--latent", type=int, default=512)
    parser.add_argument("--n_mlp", type=int, default=8)
    parser.add_argument("--device", type=str, default="cuda")
    parser.add_argument("--batch_size", type=int, default=3, help="batch size for encoding images")
    parser.add_argument("--save_all", action="store_true")
    parser.add_argument("--mixing", type=float, default=0.95)
    parser.add_argument("--smooth", type=int, default=5)
    parser.add_argument("--rotate_checkpoint", type=str, default="pretrained_models/Rotate/rotate_best.pth")
    parser.add_argument("--blending_checkpoint", type=str, default="pretrained_models/Blending/checkpoint.pth")
    parser.add_argument("--pp_checkpoint", type=str, default="pretrained_models/PostProcess/pp_model.pth")
    return parser


class HairFastHotPath(torch.nn.Module):
    """Generator + e4e + FS encoder with the per-triple call schedule of the pipeline."""

    def __init__(self, args, generator_state, e4e_state=None, fs_state=None, e4e_latent_avg=None, fs_dlatent_avg=None):
        super().__init__()
        self.args = args
        self.net = Net(args, state=generator_state)
        dev = args.device
        self.e4e = argparse.Namespace(
            encoder=Encoder4Editing(50, "ir_se", argparse.Namespace(stylegan_size=args.size)).eval(),
            opts=argparse.Namespace(start_from_latent_avg=True), latent_avg=None)
        if e4e_state is not None:
            self.e4e.encoder.load_state_dict(e4e_state)
        self.e4e.encoder.to(dev)
        self.e4e.latent_avg = (e4e_latent_avg if e4e_latent_avg is not None else torch.zeros(18, 512)).to(dev)
        self.encoder = FSEncoder(generator=self.net.generator)
        if fs_state is not None:
            self.encoder.enc.load_state_dict(fs_state)
        self.encoder.to(dev)
        if fs_dlatent_avg is not None:
            self.encoder.dlatent_avg.copy_(fs_dlatent_avg)
        self._graphs = {}

    def _call(self, key, fn, *tensors, use_graphs=False):
        """Run `fn(*tensors)` eagerly or through a per-call-site hipGraph (captured on first use)."""
        if not use_graphs:
            return fn(*tensors)
        from .graphs import GraphRunner

        k = (key,) + tuple(tuple(t.shape) for t in tensors)
        if k not in self._graphs:
            self._graphs[k] = GraphRunner(fn, *tensors)
        return self._graphs[k](*tensors)

    @torch.inference_mode()
    def swap_schedule(self, images_1024, images_256, align_inputs_256, f_align_32, f_final_64, s_blend, s_final,
                      w_rotate, include_discarded_forward=False, use_graphs=False):
        """One triple.  images_1024 / images_256: the 3 normalised inputs [3,3,1024,1024] /
        [3,3,256,256]; align_inputs_256 [2,3,256,256]: SEAN outputs re-embedded by
        Embedding.get_e4e_embed; f_align_32 [1,512,32,32], f_final_64 [1,512,64,64]: F-space
        tensors entering the last two generator calls; s_blend / s_final / w_rotate [1,18,512].
        use_graphs: replay each call site as a captured hipGraph (hairfastgan_amd/graphs.py)."""
        g = self.net.generator
        ug = use_graphs
        out = {}
        gen = lambda **kw: (lambda w_, *li: g([w_], input_is_latent=True, return_latents=False,  # noqa: E731
                                              layer_in=(li[0] if li else None), **kw)[0])
        # --- Embedding.embedding_images (Embedding.py:64-92), batch 3
        w = self._call("e4e", lambda x: get_latents(self.e4e, x), images_256, use_graphs=ug)
        self.encoder.run_discarded_generator = include_discarded_forward

        def fs(img):
            res = self.encoder.test(img=img, return_latent=True)
            return res[2], res[3]

        s, fea = self._call("fs", fs, images_1024, use_graphs=ug)
        out["F"] = self._call("g33", gen(start_layer=3, end_layer=3), s, fea, use_graphs=ug)
        out["f_from_w"] = self._call("g03", gen(start_layer=0, end_layer=3), w, use_graphs=ug)
        out["W"], out["S"] = w, s
        # --- Alignment.shape_module x2 (Alignment.py:63): full forwards of rotated latents, batch 1
        out["I_rot_shape"] = self._call("g08", gen(), w_rotate, use_graphs=ug)
        if ug:
            out["I_rot_shape"] = out["I_rot_shape"].clone()  # the same graph is replayed below
        # --- Alignment.align_images -> Embedding.get_e4e_embed (Embedding.py:44-54), batch 2
        w2 = self._call("e4e", lambda x: get_latents(self.e4e, x), align_inputs_256, use_graphs=ug)
        out["F_sean"] = self._call("g03", gen(start_layer=0, end_layer=3), w2, use_graphs=ug)
        out["I_rot_color"] = self._call("g08", gen(), w_rotate, use_graphs=ug)
        # --- Blending.blend_images (Blending.py:62, 68)
        out["I_blend"] = self._call("g48", gen(start_layer=4, end_layer=8), s_blend, f_align_32, use_graphs=ug)
        out["I_final"] = self._call("g58", gen(start_layer=5, end_layer=8), s_final, f_final_64, use_graphs=ug)
        return out
