from .model import Generator  # noqa: F401
