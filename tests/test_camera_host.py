"""Host side of nerfies_amd.camera.Camera (no GPU): constructor / JSON round trip / parameter edits behave like
nerfies/camera.py:108-180 and :323-426.  The per-pixel methods are GPU kernels: tests/test_gpu_camera.py."""
import json

import numpy as np
import pytest

from nerfies_amd.camera import Camera


def _cam(**kw):
  args = dict(orientation=np.eye(3), position=[0.1, -0.2, 0.3], focal_length=500.0, principal_point=[320.0, 240.0],
              image_size=[640, 480], skew=0.1, pixel_aspect_ratio=1.01, radial_distortion=[0.01, 0.0, 0.0],
              tangential_distortion=[0.0, 0.001])
  args.update(kw)
  return Camera(**args)


def test_properties_and_defaults():
  c = _cam(radial_distortion=None, tangential_distortion=None)
  assert not c.has_radial_distortion and not c.has_tangential_distortion
  assert c.image_shape == (480, 640) and c.orientation.dtype == np.float32 and c.image_size.dtype == np.uint32
  np.testing.assert_allclose(c.scale_factor_y, 505.0)
  np.testing.assert_allclose(c.translation, -c.position)
  np.testing.assert_array_equal(c.optical_axis, [0, 0, 1])
  d = _cam()
  assert d.has_radial_distortion and d.has_tangential_distortion
  px = d.get_pixel_centers()
  assert px.shape == (480, 640, 2) and px[0, 0].tolist() == [0.5, 0.5] and px[-1, -1].tolist() == [639.5, 479.5]


def test_json_round_trip(tmp_path):
  c = _cam()
  path = tmp_path / 'cam.json'
  path.write_text(json.dumps(c.to_json()))
  d = Camera.from_json(str(path))
  for k, v in c.get_parameters().items():
    np.testing.assert_allclose(np.asarray(d.get_parameters()[k], np.float64), np.asarray(v, np.float64), rtol=1e-6)
  old = c.to_json()
  old['tangential'] = old.pop('tangential_distortion')     # legacy key (camera.py:150-152)
  old['tangential_distortion'] = [0.0, 0.0]
  path.write_text(json.dumps(old))
  np.testing.assert_allclose(Camera.from_json(str(path)).tangential_distortion, c.tangential_distortion)


def test_scale_crop_look_at():
  c = _cam()
  s = c.scale(0.5)
  assert s.image_size.tolist() == [320, 240] and float(s.focal_length) == 250.0 and s.principal_point.tolist() == [160.0, 120.0]
  with pytest.raises(ValueError):
    c.scale(0.0)
  k = c.crop_image_domain(left=10, right=20, top=5, bottom=-5)
  assert k.image_size.tolist() == [610, 480] and k.principal_point.tolist() == [310.0, 235.0]
  with pytest.raises(ValueError):
    c.crop_image_domain(left=400, right=400)
  la = c.look_at(np.array([0.0, 0.0, -2.0]), np.array([0.0, 0.0, 0.0]), np.array([0.0, 1.0, 0.0]))
  R = np.asarray(la.orientation, np.float64)
  np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-12)
  np.testing.assert_allclose(R[2], [0, 0, 1], atol=1e-12)
  assert np.linalg.det(R) > 0
  with pytest.raises(ValueError):
    c.look_at(np.zeros(3), np.zeros(3), np.array([0.0, 1.0, 0.0]))
  with pytest.raises(ValueError):
    c.look_at(np.zeros(3), np.array([0.0, 1.0, 0.0]), np.array([0.0, 1.0, 0.0]))
  assert c.copy() is not c and c.copy().focal_length == c.focal_length
