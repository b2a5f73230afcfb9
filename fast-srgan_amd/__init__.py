"""fast-srgan_amd: MI355X-native (gfx950) hot path of Fast-SRGAN behind the reference's Python surface.

The directory name carries a hyphen (it mirrors the upstream repository name), so import it with
`importlib.import_module("fast-srgan_amd")` or through the `fast_srgan_amd` alias module at the repo root.
"""
from . import _lib  # noqa: F401
from .config import load_config  # noqa: F401
from .dataloader import DeviceBatchLoader, NumpyImagesDataset  # noqa: F401
from .inference import InferencePipeline  # noqa: F401
from .model import VGG19, Discriminator, Generator, GraphedGenerator  # noqa: F401
from .optim import ArenaAdamW  # noqa: F401
from .trainer import Trainer  # noqa: F401

__all__ = ["Generator", "GraphedGenerator", "Discriminator", "VGG19", "Trainer", "NumpyImagesDataset", "DeviceBatchLoader", "ArenaAdamW",
           "load_config", "InferencePipeline"]
