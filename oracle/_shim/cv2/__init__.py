"""cv2 -> PIL for the one call the dataset reader makes: imdecode of a PNG into a BGR uint8 array
(nerfies/datasets/nerfies.py:55-61 then flips it to RGB)."""
import io as _io

import numpy as _np
from PIL import Image as _Image

IMREAD_COLOR = 1
INTER_AREA = 3


def imdecode(buf, flags):
  rgb = _np.asarray(_Image.open(_io.BytesIO(_np.asarray(buf, _np.uint8).tobytes())).convert('RGB'))
  return rgb[:, :, ::-1].copy()


def resize(*a, **k):
  raise NotImplementedError('cv2.resize is outside the shim')
