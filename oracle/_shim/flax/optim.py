class Optimizer:
  pass


class Adam:
  def __init__(self, *a, **k):
    pass
