"""MI355X-native implementation of BAGEL's image-generation hot path."""
from .config import BagelDims, VAEDims, BAGEL_7B, BAGEL_0P5B  # noqa: F401
