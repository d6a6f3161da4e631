from .bagel import Bagel, NaiveCache  # noqa: F401
from .autoencoder import AutoEncoder  # noqa: F401
