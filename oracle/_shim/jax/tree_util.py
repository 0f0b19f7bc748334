def tree_map(f, tree, *rest):
  if isinstance(tree, dict):
    return {k: tree_map(f, v, *[r[k] for r in rest]) for k, v in tree.items()}
  if isinstance(tree, (list, tuple)):
    return type(tree)(tree_map(f, v, *[r[i] for r in rest]) for i, v in enumerate(tree))
  return f(tree, *rest)


tree_multimap = tree_map
