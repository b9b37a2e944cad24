from .augmentations import *  # noqa: F401,F403
