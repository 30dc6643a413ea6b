"""MI355X-native TecoGAN training hot path (generator + pseudo-flow/warp + spatio-temporal discriminator + losses +
Adam) behind the reference's Python surface.  The directory name is not a Python identifier, so the package is
imported as `pytorch_tecogan_amd` through the loader module of that name at the repository root."""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
