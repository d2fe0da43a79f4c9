"""Encoder forwards that feed the generator and the
PostProcess encoder in front of its last call (row f1); the ModulationModule stacks of row f4 (RotateModel,
ClipBlendingModel around an injected CLIP image tower)."""
from .e4e import Encoder4Editing, get_latents  # noqa: F401
from .fs_encoder import FSEncoder, fs_encoder_v2  # noqa: F401
from .post_process import ClipBlendingModel, PostProcessModel, RotateModel  # noqa: F401
