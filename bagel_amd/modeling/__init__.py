from .bagel import Bagel, NaiveCache  # noqa: F401
