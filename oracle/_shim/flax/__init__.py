from . import jax_utils, linen, optim, struct  # noqa: F401
