"""GPU parity of the camera kernels (nrf_camera_*, csrc/camera.hip) against oracle/camera_oracle.py (fp64, pinned
to the reference's Camera class by tests/test_reference_vectors.py) and against the reference's own vectors.
Tolerances: the kernels compute in fp32 like the reference's default camera dtype; unit directions agree to 2e-6,
pixel positions to 2e-3 px at ~500 px focal length (fp32 eps * coordinate magnitude * a few ops)."""
import os

import numpy as np
import pytest
import torch

from nerfies_amd.camera import Camera, camera_to_rays
from oracle import camera_oracle as CO

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

DIST = dict(radial_distortion=[0.05, -0.02, 0.004], tangential_distortion=[0.001, -0.002])


def _pair(seed=0, size=(320, 240), focal=412.5, distorted=True, skew=0.3, par=1.02):
  rng = np.random.default_rng(seed)
  R, _ = np.linalg.qr(rng.normal(size=(3, 3)))
  kw = dict(orientation=R, position=rng.normal(size=3), focal_length=focal, principal_point=[size[0] / 2 + 0.2, size[1] / 2 - 0.3],
            image_size=list(size), skew=skew, pixel_aspect_ratio=par)
  if distorted:
    kw.update(DIST)
  cam = Camera(**kw)
  ocam = CO.make_camera(cam.orientation, cam.position, cam.focal_length, cam.principal_point, size, cam.skew,
                        cam.pixel_aspect_ratio, cam.radial_distortion, cam.tangential_distortion)
  return cam, ocam, rng


@pytest.mark.parametrize('distorted', [False, True])
@pytest.mark.parametrize('n', [1, 255, 256, 1000, 70001])
def test_pixels_to_rays(n, distorted):
  cam, ocam, rng = _pair(n, distorted=distorted)
  px = rng.uniform(0, [320, 240], size=(n, 2)).astype(np.float32)
  got = cam.pixels_to_rays(torch.from_numpy(px).cuda())
  assert got.is_cuda and got.shape == (n, 3)
  want = CO.pixels_to_rays(ocam, px)
  np.testing.assert_allclose(got.cpu().numpy(), want, rtol=0, atol=2e-6)
  # numpy in -> numpy out, batch shape kept
  if n == 1000:
    got_np = cam.pixels_to_rays(px.reshape(10, 100, 2))
    assert isinstance(got_np, np.ndarray) and got_np.shape == (10, 100, 3)
    np.testing.assert_array_equal(got_np.reshape(-1, 3), got.cpu().numpy())


@pytest.mark.parametrize('tag', ['pinhole', 'distorted'])
def test_against_reference_vectors(tag):
  r = np.load(os.path.join(HERE, 'golden', 'ref_camera.npz'))
  f, cx, cy, skew, par = r['intrinsics']
  cam = Camera(orientation=r[f'{tag}/orientation'], position=r[f'{tag}/position'], focal_length=f, principal_point=[cx, cy],
               image_size=[320, 240], skew=skew, pixel_aspect_ratio=par, radial_distortion=r[f'{tag}/radial'],
               tangential_distortion=r[f'{tag}/tangential'])
  np.testing.assert_allclose(cam.pixels_to_rays(r[f'{tag}/pixels'].astype(np.float32)), r[f'{tag}/rays'], rtol=0, atol=2e-6)
  np.testing.assert_allclose(cam.project(r[f'{tag}/points'].astype(np.float32)), r[f'{tag}/project'], rtol=0, atol=2e-3)
  small = Camera(orientation=r[f'{tag}/orientation'], position=r[f'{tag}/position'], focal_length=20.0,
                 principal_point=[3.5, 2.5], image_size=[7, 5], radial_distortion=r[f'{tag}/radial'],
                 tangential_distortion=r[f'{tag}/tangential'])
  rays = camera_to_rays(small)
  np.testing.assert_array_equal(rays['pixels'].cpu().numpy(), r[f'{tag}/centers_7x5'].astype(np.float32))
  np.testing.assert_allclose(rays['directions'].cpu().numpy(), r[f'{tag}/centers_rays_7x5'], rtol=0, atol=2e-6)


@pytest.mark.parametrize('distorted', [False, True])
def test_camera_to_rays_full_frame(distorted):
  """datasets/core.py:50-75 on a 960x540 frame: shapes, dtypes, tiled origins, pixel centres, oracle directions."""
  cam, ocam, _ = _pair(3, size=(960, 540), focal=800.0, distorted=distorted)
  rays = camera_to_rays(cam)
  want = CO.camera_to_rays(ocam)
  for k in ('origins', 'directions', 'pixels'):
    assert rays[k].dtype == torch.float32 and tuple(rays[k].shape) == want[k].shape
  np.testing.assert_array_equal(rays['origins'].cpu().numpy(), want['origins'])
  np.testing.assert_array_equal(rays['pixels'].cpu().numpy(), want['pixels'])
  np.testing.assert_allclose(rays['directions'].cpu().numpy(), want['directions'], rtol=0, atol=2e-6)
  norms = rays['directions'].norm(dim=-1)
  assert float((norms - 1).abs().max()) < 1e-6


@pytest.mark.parametrize('distorted', [False, True])
def test_project_and_round_trip(distorted):
  cam, ocam, rng = _pair(5, distorted=distorted)
  n = 5000
  px = rng.uniform(0, [320, 240], size=(n, 2)).astype(np.float32)
  depth = rng.uniform(0.5, 3.0, n).astype(np.float32)
  pts = cam.pixels_to_points(torch.from_numpy(px).cuda(), torch.from_numpy(depth).cuda())
  np.testing.assert_allclose(pts.cpu().numpy(), CO.pixels_to_points(ocam, px, depth), rtol=0, atol=2e-5)
  # depth is measured along the optical axis
  local_z = (pts.cpu().numpy().astype(np.float64) - ocam['position']) @ ocam['orientation'][2]
  np.testing.assert_allclose(local_z, depth, rtol=2e-5)
  back = cam.project(pts)
  np.testing.assert_allclose(back.cpu().numpy(), CO.project(ocam, pts.cpu().numpy()), rtol=0, atol=2e-3)
  np.testing.assert_allclose(back.cpu().numpy(), px, rtol=0, atol=5e-3)     # undistort o distort = id


def test_errors():
  from nerfies_amd import lib as L
  cam, _, _ = _pair(0)
  with pytest.raises(ValueError):
    cam.pixels_to_rays(torch.zeros(4, 3, device='cuda'))
  with pytest.raises(ValueError):
    cam.pixels_to_rays(torch.zeros(4, 2, device='cuda', dtype=torch.float64))
  assert cam.pixels_to_rays(torch.zeros(0, 2, device='cuda')).shape == (0, 3)
  lib = L.load_library()
  d = cam._desc()
  out = torch.empty(12, device='cuda')
  assert lib.nrf_camera_pixels_to_rays(d, None, 4, None, out.data_ptr(), None, None) != 0     # n != W*H
  assert b'width*height' in lib.nrf_last_error()
  d.focal_length = 0.0
  assert lib.nrf_camera_project(d, out.data_ptr(), 4, out.data_ptr(), None) != 0


def test_render_frame_from_camera_matches_oracle():
  """Camera -> rays on the GPU -> chunked render (evaluation.py:62-99) == the oracle applied to oracle-camera rays."""
  import helpers as H
  from nerfies_amd import evaluation, training
  from oracle import nerfies_oracle as O
  spec = O.ModelSpec(num_coarse_samples=16, num_fine_samples=16, num_nerf_point_freqs=4, use_stratified_sampling=False,
                     use_warp=True, num_warp_freqs=4, use_camera_metadata=True)
  oparams = O.init_params(spec, seed=3, trained_like=True)
  model, fp = H.gpu_model(spec, oparams)
  cam, ocam, _ = _pair(9, size=(12, 9), focal=15.0, skew=0.0, par=1.0)
  cam.position[:] = [0.05, -0.02, 0.1]; ocam['position'][:] = cam.position
  meta = {'warp': 2, 'camera': 1, 'appearance': 0}
  rays = evaluation.rays_from_camera(cam, meta)
  assert rays['metadata']['warp'].shape == (9, 12, 1) and rays['metadata']['warp'].dtype == torch.int32
  state = training.TrainState(optimizer=training.Optimizer(fp), warp_alpha=2.0)
  fn = lambda k0, k1, params, r, extra: model.apply({'params': params}, r, extra)
  img = evaluation.render_image(state, rays, fn, 1, 0, chunk=50)
  want_rays = CO.camera_to_rays(ocam)
  n = 12 * 9
  batch = {'origins': torch.from_numpy(want_rays['origins'].reshape(n, 3)).double(),
           'directions': torch.from_numpy(want_rays['directions'].reshape(n, 3)).double(),
           'metadata': {k: torch.full((n, 1), v, dtype=torch.int64) for k, v in meta.items()}}
  ref = O.nerf_model_apply(oparams, spec, batch, 2.0)
  got = img['rgb'].reshape(n, 3).cpu().numpy()
  np.testing.assert_allclose(got, ref['fine']['rgb'].numpy(), rtol=0, atol=2e-4)
  np.testing.assert_allclose(img['depth'].reshape(n).cpu().numpy(), ref['fine']['depth'].numpy(), rtol=0, atol=2e-4)
  m = evaluation.image_metrics(img['rgb'], torch.from_numpy(ref['fine']['rgb'].numpy()).float().reshape(9, 12, 3))
  assert float(m['psnr']) > 60
