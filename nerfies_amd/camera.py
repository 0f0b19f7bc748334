"""Camera geometry with the per-pixel work on the GPU.

Mirror of nerfies/camera.py `Camera` (same constructor, JSON format, properties and method names).  The per-pixel
methods -- pixels_to_rays (with the 10-step Newton undistort), pixels_to_points, project -- run as HIP kernels
through the C-ABI (nrf_camera_*, csrc/camera.hip); `camera_to_rays` (datasets/core.py:50-75) renders the origins /
directions / pixels of a whole frame in one launch, directly into the device tensors the renderer consumes, instead
of the reference's per-frame host NumPy pass.  The remaining methods only edit the nine camera parameters on the
host, as in the reference.

Arrays: a CUDA torch tensor in -> a CUDA tensor out (no copy); a NumPy array in -> uploaded, NumPy out.
There is no CPU implementation: without the HIP library these methods raise."""
import copy
import json

import numpy as np
import torch

from . import lib as L


def _stream():
  return torch.cuda.current_stream().cuda_stream


class Camera:
  """Pinhole camera with skew, pixel aspect ratio and Brown-Conrady distortion (camera.py:108-140)."""

  def __init__(self, orientation, position, focal_length, principal_point, image_size, skew=0.0,
               pixel_aspect_ratio=1.0, radial_distortion=None, tangential_distortion=None, dtype=np.float32):
    if radial_distortion is None:
      radial_distortion = np.zeros(3, dtype)
    if tangential_distortion is None:
      tangential_distortion = np.zeros(2, dtype)
    self.orientation = np.array(orientation, dtype)
    self.position = np.array(position, dtype)
    self.focal_length = np.array(focal_length, dtype)
    self.principal_point = np.array(principal_point, dtype)
    self.skew = np.array(skew, dtype)
    self.pixel_aspect_ratio = np.array(pixel_aspect_ratio, dtype)
    self.radial_distortion = np.array(radial_distortion, dtype)
    self.tangential_distortion = np.array(tangential_distortion, dtype)
    self.image_size = np.array(image_size, np.uint32)
    self.dtype = dtype

  # ---- (de)serialisation: camera.py:142-180 ----
  @classmethod
  def from_json(cls, path):
    with open(path, 'r') as fp:
      cj = json.load(fp)
    if 'tangential' in cj:                      # old camera files
      cj['tangential_distortion'] = cj['tangential']
    return cls(orientation=np.asarray(cj['orientation']), position=np.asarray(cj['position']),
               focal_length=cj['focal_length'], principal_point=np.asarray(cj['principal_point']), skew=cj['skew'],
               pixel_aspect_ratio=cj['pixel_aspect_ratio'], radial_distortion=np.asarray(cj['radial_distortion']),
               tangential_distortion=np.asarray(cj['tangential_distortion']), image_size=np.asarray(cj['image_size']))

  def get_parameters(self):
    return {k: getattr(self, k) for k in (
        'orientation', 'position', 'focal_length', 'principal_point', 'skew', 'pixel_aspect_ratio',
        'radial_distortion', 'tangential_distortion', 'image_size')}

  def to_json(self):
    return {k: (v.tolist() if hasattr(v, 'tolist') else v) for k, v in self.get_parameters().items()}

  # ---- derived quantities: camera.py:182-223 ----
  scale_factor_x = property(lambda self: self.focal_length)
  scale_factor_y = property(lambda self: self.focal_length * self.pixel_aspect_ratio)
  principal_point_x = property(lambda self: self.principal_point[0])
  principal_point_y = property(lambda self: self.principal_point[1])
  has_tangential_distortion = property(lambda self: bool(np.any(self.tangential_distortion != 0.0)))
  has_radial_distortion = property(lambda self: bool(np.any(self.radial_distortion != 0.0)))
  image_size_x = property(lambda self: self.image_size[0])
  image_size_y = property(lambda self: self.image_size[1])
  image_shape = property(lambda self: (self.image_size_y, self.image_size_x))
  optical_axis = property(lambda self: self.orientation[2, :])
  translation = property(lambda self: -np.matmul(self.orientation, self.position))

  # ---- device side ----
  def _desc(self):
    d = L.CameraDesc()
    d.orientation[:] = [float(v) for v in np.asarray(self.orientation, np.float64).reshape(9)]
    d.position[:] = [float(v) for v in self.position]
    d.focal_length = float(self.focal_length)
    d.principal_point[:] = [float(v) for v in self.principal_point]
    d.skew = float(self.skew)
    d.pixel_aspect_ratio = float(self.pixel_aspect_ratio)
    d.radial_distortion[:] = [float(v) for v in self.radial_distortion]
    d.tangential_distortion[:] = [float(v) for v in self.tangential_distortion]
    d.image_size[:] = [int(self.image_size[0]), int(self.image_size[1])]
    return d

  @staticmethod
  def _to_device(x, last, device):
    """-> (flat float32 CUDA tensor [n, last], batch shape, was_numpy)."""
    is_np = not torch.is_tensor(x)
    t = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)) if is_np else x
    if t.shape[-1] != last:
      raise ValueError(f'The last dimension must be {last}.')
    if t.dtype != torch.float32:
      raise ValueError(f'dtype ({t.dtype}) must be float32: the kernels compute in fp32')
    if not t.is_cuda:
      t = t.to(device or 'cuda')
    return t.reshape(-1, last).contiguous(), tuple(t.shape[:-1]), is_np

  @staticmethod
  def _back(t, is_np):
    return t.cpu().numpy() if is_np else t

  def pixels_to_rays(self, pixels, device=None):
    """[..., 2] pixel positions -> [..., 3] unit ray directions in world space (camera.py:244-269)."""
    px, batch, is_np = self._to_device(pixels, 2, device)
    lib = L.load_library()
    out = torch.empty((px.shape[0], 3), dtype=torch.float32, device=px.device)
    if px.shape[0]:
      with torch.cuda.device(px.device):
        L.check(lib.nrf_camera_pixels_to_rays(self._desc(), px.data_ptr(), px.shape[0], None, out.data_ptr(), None,
                                              _stream()), lib)
    return self._back(out.reshape(batch + (3,)), is_np)

  def pixels_to_points(self, pixels, depth, device=None):
    """Points at `depth` along the optical axis through each pixel (camera.py:271-277)."""
    px, batch, is_np = self._to_device(pixels, 2, device)
    dp = torch.as_tensor(np.ascontiguousarray(depth, dtype=np.float32)) if not torch.is_tensor(depth) else depth
    dp = dp.to(px.device, torch.float32).reshape(-1).contiguous()
    if dp.shape[0] != px.shape[0]:
      raise ValueError('depth must have one value per pixel')
    lib = L.load_library()
    out = torch.empty((px.shape[0], 3), dtype=torch.float32, device=px.device)
    if px.shape[0]:
      with torch.cuda.device(px.device):
        L.check(lib.nrf_camera_pixels_to_points(self._desc(), px.data_ptr(), dp.data_ptr(), px.shape[0],
                                                out.data_ptr(), _stream()), lib)
    return self._back(out.reshape(batch + (3,)), is_np)

  def project(self, points, device=None):
    """[..., 3] world points -> [..., 2] (distorted) pixel positions (camera.py:283-315)."""
    pts, batch, is_np = self._to_device(points, 3, device)
    lib = L.load_library()
    out = torch.empty((pts.shape[0], 2), dtype=torch.float32, device=pts.device)
    if pts.shape[0]:
      with torch.cuda.device(pts.device):
        L.check(lib.nrf_camera_project(self._desc(), pts.data_ptr(), pts.shape[0], out.data_ptr(), _stream()), lib)
    return self._back(out.reshape(batch + (2,)), is_np)

  def get_pixel_centers(self):
    """[H, W, 2] pixel centres (x + 0.5, y + 0.5) (camera.py:317-321)."""
    xx, yy = np.meshgrid(np.arange(self.image_size_x, dtype=self.dtype), np.arange(self.image_size_y, dtype=self.dtype))
    return np.stack([xx, yy], axis=-1) + 0.5

  def to_rays(self, device='cuda'):
    """datasets/core.py:50-75 camera_to_rays as one launch: {'origins','directions','pixels'} [H, W, .] on `device`."""
    h, w = int(self.image_size_y), int(self.image_size_x)
    dev = torch.device(device)
    origins = torch.empty((h, w, 3), dtype=torch.float32, device=dev)
    directions = torch.empty((h, w, 3), dtype=torch.float32, device=dev)
    pixels = torch.empty((h, w, 2), dtype=torch.float32, device=dev)
    lib = L.load_library()
    with torch.cuda.device(dev):
      L.check(lib.nrf_camera_pixels_to_rays(self._desc(), None, h * w, origins.data_ptr(), directions.data_ptr(),
                                            pixels.data_ptr(), _stream()), lib)
    return {'origins': origins, 'directions': directions, 'pixels': pixels}

  # ---- host-side parameter edits: camera.py:323-426 ----
  def scale(self, scale):
    if scale <= 0:
      raise ValueError('scale needs to be positive.')
    return Camera(orientation=self.orientation.copy(), position=self.position.copy(),
                  focal_length=self.focal_length * scale, principal_point=self.principal_point.copy() * scale,
                  skew=self.skew, pixel_aspect_ratio=self.pixel_aspect_ratio,
                  radial_distortion=self.radial_distortion.copy(),
                  tangential_distortion=self.tangential_distortion.copy(),
                  image_size=np.array((int(round(self.image_size[0] * scale)), int(round(self.image_size[1] * scale)))))

  def look_at(self, position, look_at, up, eps=1e-6):
    """Copy of this camera at `position`, optical axis through `look_at`, image y axis along the projection of `up`."""
    axis = np.asarray(look_at, np.float64) - np.asarray(position, np.float64)
    n = np.linalg.norm(axis)
    if n < eps:
      raise ValueError('The camera center and look at position are too close.')
    axis = axis / n
    right = np.cross(axis, up)
    n = np.linalg.norm(right)
    if n < eps:
      raise ValueError('The up-vector is parallel to the optical axis.')
    right = right / n
    cam = self.copy()
    cam.position = position
    cam.orientation = np.stack([right, np.cross(axis, right), axis], axis=0)
    return cam

  def crop_image_domain(self, left=0, right=0, top=0, bottom=0):
    """Copy with the image bounds moved inwards (negative: outwards); the principal axis is preserved."""
    lt, rb = np.array([left, top]), np.array([right, bottom])
    size = self.image_size - lt - rb
    if np.any(size <= 0):
      raise ValueError('Crop would result in non-positive image dimensions.')
    cam = self.copy()
    cam.image_size = np.array([int(size[0]), int(size[1])])
    pp = self.principal_point - lt
    cam.principal_point = np.array([pp[0], pp[1]])
    return cam

  def copy(self):
    return copy.deepcopy(self)


def camera_to_rays(camera, device='cuda'):
  """datasets/core.py:50-75."""
  return camera.to_rays(device)
