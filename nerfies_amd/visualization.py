"""Colour-mapping of scalar images for the evaluation renders (nerfies/visualization.py:150-219) and the 8/16-bit
image writers of nerfies/image_utils.py:108-173 (PIL)."""
import functools

import numpy as np


@functools.lru_cache(maxsize=32)
def get_colormap(name='magma', num_bins=256):
  """[num_bins, 3] RGB table of a matplotlib colormap (visualization.py:162-176)."""
  import matplotlib
  base = matplotlib.colormaps[name] if hasattr(matplotlib, 'colormaps') else matplotlib.cm.get_cmap(name)
  return np.asarray(base(np.linspace(0, 1, num_bins)))[:, :3]


def colorize(array, cmin=None, cmax=None, cmap='magma', eps=1e-6, invert=False):
  """Scalar image -> RGB via a 256-entry colour table with linear interpolation; values outside [cmin, cmax]
  saturate to black / white (visualization.py:191-219)."""
  array = np.asarray(array, np.float64)
  cmin = array.min() if cmin is None else cmin
  cmax = array.max() if cmax is None else cmax
  x = (array - cmin) / max(cmax - cmin, eps)
  table = get_colormap(cmap)
  v = np.clip(1.0 - x if invert else x, 0.0, 1.0) * 255.0
  lo = np.floor(v).astype(np.int64)
  hi = np.minimum(lo + 1, 255)
  out = table[lo] + (table[hi] - table[lo]) * (v - lo)[..., None]
  out[x > 1.0] = 0.0 if invert else 1.0
  out[x < 0.0] = 1.0 if invert else 0.0
  return out


def image_to_uint8(image):
  image = np.asarray(image)
  if image.dtype == np.uint8:
    return image
  if not np.issubdtype(image.dtype, np.floating):
    raise ValueError(f'Input image should be a floating type but is of type {image.dtype!r}')
  return (image * 255).clip(0.0, 255).astype(np.uint8)


def image_to_uint16(image):
  image = np.asarray(image)
  if image.dtype == np.uint16:
    return image
  if not np.issubdtype(image.dtype, np.floating):
    raise ValueError(f'Input image should be a floating type but is of type {image.dtype!r}')
  return (image * 65535).clip(0.0, 65535).astype(np.uint16)


def save_image(path, image):
  from PIL import Image
  Image.fromarray(np.asarray(image)).save(path)


def save_depth(path, depth):
  """16-bit PNG of depth / 1000 (image_utils.py:164-165)."""
  save_image(path, image_to_uint16(np.asarray(depth, np.float64) / 1000.0))
