"""imageio: imported by nerfies.image_utils; not used by the paths the shim drives."""


def imread(*a, **k):
  raise NotImplementedError('imageio.imread is outside the shim')
