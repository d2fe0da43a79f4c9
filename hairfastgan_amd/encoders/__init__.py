"""Encoder forwards that feed the generator."""
from .e4e import Encoder4Editing, get_latents  # noqa: F401
from .fs_encoder import FSEncoder, fs_encoder_v2  # noqa: F401
