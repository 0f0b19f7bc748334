from . import numpy  # noqa: F401
