def stop_gradient(x):
  return x


def pmean(x, axis_name=None):
  return x


def all_gather(x, axis_name=None):
  return x
