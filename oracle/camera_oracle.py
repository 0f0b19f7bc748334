"""CPU oracle for the camera geometry (NumPy float64).  TEST INFRASTRUCTURE ONLY: imported by tests/ and
__graft_entry__.smoke(); the product (nerfies_amd/camera.py -> nrf_camera_* HIP kernels) never imports it.

Restates nerfies/camera.py: undistort :26-105, pixel_to_local_rays :225-242, pixels_to_rays :244-269,
pixels_to_points :271-277, project :283-315, get_pixel_centers :317-321, and datasets/core.py:50-75
camera_to_rays.  PINNED: tests/test_reference_vectors.py checks it against tests/golden/ref_camera.npz, produced
by the reference's own Camera class (tests/golden/make_reference_vectors.py)."""
import numpy as np


def make_camera(orientation, position, focal_length, principal_point, image_size, skew=0.0, pixel_aspect_ratio=1.0,
                radial_distortion=(0.0, 0.0, 0.0), tangential_distortion=(0.0, 0.0)):
  f64 = lambda a: np.asarray(a, np.float64)
  return dict(orientation=f64(orientation).reshape(3, 3), position=f64(position), focal_length=float(focal_length),
              principal_point=f64(principal_point), image_size=tuple(int(v) for v in image_size), skew=float(skew),
              pixel_aspect_ratio=float(pixel_aspect_ratio), radial_distortion=f64(radial_distortion),
              tangential_distortion=f64(tangential_distortion))


def distort(x, y, k, p):
  """Brown-Conrady model in normalised image coordinates (camera.py:293-308)."""
  r2 = x * x + y * y
  d = 1.0 + r2 * (k[0] + r2 * (k[1] + k[2] * r2))
  return (x * d + 2.0 * p[0] * x * y + p[1] * (r2 + 2.0 * x * x),
          y * d + 2.0 * p[1] * x * y + p[0] * (r2 + 2.0 * y * y))


def undistort(xd, yd, k, p, eps=1e-9, iterations=10):
  """Fixed-count Newton iteration on distort(x, y) = (xd, yd), started at the distorted point; where the
  Jacobian determinant is within eps of zero the point is left alone for that step (camera.py:76-105)."""
  x, y = np.array(xd, np.float64), np.array(yd, np.float64)
  for _ in range(iterations):
    r2 = x * x + y * y
    d = 1.0 + r2 * (k[0] + r2 * (k[1] + k[2] * r2))
    gx, gy = distort(x, y, k, p)
    fx, fy = gx - xd, gy - yd
    dd = k[0] + r2 * (2.0 * k[1] + 3.0 * k[2] * r2)       # d'(r2)
    a = d + 2.0 * x * x * dd + 2.0 * p[0] * y + 6.0 * p[1] * x      # dfx/dx
    b = 2.0 * x * y * dd + 2.0 * p[0] * x + 2.0 * p[1] * y          # dfx/dy
    c = 2.0 * x * y * dd + 2.0 * p[1] * y + 2.0 * p[0] * x          # dfy/dx
    e = d + 2.0 * y * y * dd + 2.0 * p[1] * x + 6.0 * p[0] * y      # dfy/dy
    det = c * b - a * e                                             # = -(det J)
    ok = np.abs(det) > eps
    safe = np.where(ok, det, 1.0)
    x = x + np.where(ok, (fx * e - fy * b) / safe, 0.0)
    y = y + np.where(ok, (fy * a - fx * c) / safe, 0.0)
  return x, y


def pixel_centers(cam):
  w, h = cam['image_size']
  xx, yy = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
  return np.stack([xx, yy], -1) + 0.5


def pixels_to_rays(cam, pixels):
  pixels = np.asarray(pixels, np.float64)
  flat = pixels.reshape(-1, 2)
  y = (flat[:, 1] - cam['principal_point'][1]) / (cam['focal_length'] * cam['pixel_aspect_ratio'])
  x = (flat[:, 0] - cam['principal_point'][0] - y * cam['skew']) / cam['focal_length']
  k, p = cam['radial_distortion'], cam['tangential_distortion']
  if np.any(k != 0.0) or np.any(p != 0.0):
    x, y = undistort(x, y, k, p)
  local = np.stack([x, y, np.ones_like(x)], -1)
  local /= np.linalg.norm(local, axis=-1, keepdims=True)
  world = local @ cam['orientation']          # = (R^T local) per row
  world /= np.linalg.norm(world, axis=-1, keepdims=True)
  return world.reshape(pixels.shape[:-1] + (3,))


def pixels_to_points(cam, pixels, depth):
  rays = pixels_to_rays(cam, pixels)
  cosa = rays @ cam['orientation'][2]
  return rays * (np.asarray(depth, np.float64) / cosa)[..., None] + cam['position']


def project(cam, points):
  points = np.asarray(points, np.float64)
  local = (points.reshape(-1, 3) - cam['position']) @ cam['orientation'].T
  x, y = distort(local[:, 0] / local[:, 2], local[:, 1] / local[:, 2], cam['radial_distortion'],
                 cam['tangential_distortion'])
  px = cam['focal_length'] * x + cam['skew'] * y + cam['principal_point'][0]
  py = cam['focal_length'] * cam['pixel_aspect_ratio'] * y + cam['principal_point'][1]
  return np.stack([px, py], -1).reshape(points.shape[:-1] + (2,))


def camera_to_rays(cam):
  """datasets/core.py:50-75: origins / directions / pixels for every pixel centre, [H, W, .] float32."""
  px = pixel_centers(cam)
  return dict(origins=np.broadcast_to(cam['position'], px.shape[:-1] + (3,)).astype(np.float32),
              directions=pixels_to_rays(cam, px).astype(np.float32), pixels=px.astype(np.float32))
