"""`Net`: owner of the generator, the module boundary HairFast's stages talk to
(`self.net.generator(...)`, `self.net.latent_avg`).  Mirror of models/Net.py:20-46.

In scope: building `Generator(opts.size, opts.latent, opts.n_mlp, channel_multiplier)`,
the checkpoint contract (`ckpt['g_ema']`, `ckpt['latent_avg']`, models/Net.py:37-42),
freezing the parameters and `cal_layer_num` (:78-88).  Out of scope:
the PCA model used only by training losses (:48-76), the BiSeNet singleton trigger (:29)
and the gdown auto-download (:32-34) - a missing checkpoint raises instead.
"""
import torch
from torch import nn

from .checkpoints import load_file
from .stylegan2.model import Generator


class Net(nn.Module):
    def __init__(self, opts, state=None, root=None):
        """`state`: optional in-memory checkpoint dict {'g_ema': ..., 'latent_avg': ...}
        (used by tests / benchmarks with synthetic weights); otherwise `opts.ckpt` is loaded
        (relative to `root` / HAIRFAST_PRETRAINED_ROOT / the working directory)."""
        super().__init__()
        self.opts = opts
        self.generator = Generator(opts.size, opts.latent, opts.n_mlp, channel_multiplier=opts.channel_multiplier)
        self.cal_layer_num()
        self.load_weights(state, root)

    def load_weights(self, state=None, root=None):
        if state is None:
            state = load_file(self.opts.ckpt, "StyleGAN2 generator (--ckpt, models/Net.py:31-40)", root)
        device = self.opts.device
        self.generator.load_state_dict(state["g_ema"])
        self.latent_avg = state["latent_avg"].to(device)
        self.generator.to(device)
        for param in self.generator.parameters():
            param.requires_grad = False
        self.generator.eval()

    def cal_layer_num(self):
        self.layer_num = {1024: 18, 512: 16, 256: 14}[self.opts.size]
        self.S_index = self.layer_num - 11
