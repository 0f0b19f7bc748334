def save_checkpoint(*a, **k):
  raise NotImplementedError


def restore_checkpoint(*a, **k):
  raise NotImplementedError
