def info(*a, **k):
  pass


warning = error = debug = log = info


INFO = 20


def log_every_n_seconds(*a, **k):
  pass
