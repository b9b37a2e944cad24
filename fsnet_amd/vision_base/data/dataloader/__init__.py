from .dataloader_builder import build_dataloader  # noqa: F401
