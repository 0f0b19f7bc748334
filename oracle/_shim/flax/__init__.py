from . import linen, optim, struct  # noqa: F401
