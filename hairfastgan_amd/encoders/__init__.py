"""Encoder forwards that feed the generator and the
PostProcess encoder in front of its last call (row f1)."""
from .e4e import Encoder4Editing, get_latents  # noqa: F401
from .fs_encoder import FSEncoder, fs_encoder_v2  # noqa: F401
from .post_process import PostProcessModel  # noqa: F401
