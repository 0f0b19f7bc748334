"""flax.jax_utils -> the single-process meaning of (un)replicate."""


def replicate(tree, devices=None):
  return tree


def unreplicate(tree):
  """First replica of a pmap output: index 0 of the leading axis of every leaf."""
  if isinstance(tree, dict):
    return {k: unreplicate(v) for k, v in tree.items()}
  return tree[0]
