from .bagel import Bagel, NaiveCache  # noqa: F401
from .autoencoder import AutoEncoder  # noqa: F401
from .qwen2_navit import BaseNavitOutputWithPast, Qwen2ForCausalLM  # noqa: F401
