"""The oracle against vectors produced by the REAL reference sources (tests/golden/ref_*.npz, written by
tests/golden/make_reference_vectors.py, which imports google/nerfies from /root/reference and executes it on
NumPy float64 through the import-name stand-ins of oracle/_shim).  This is what pins oracle/nerfies_oracle.py to
the reference's own code for: rigid_body, model_utils (sampling, compositing, PDF, depth), the sinusoidal /
annealed encoders, MLP / NerfMLP, SE3Field (+ its Jacobian by finite differences of the reference warp),
NerfModel.apply end to end (coarse + fine, conditions, warp), the elastic loss, general_loss, psnr, schedules.
Parameters are regenerated from the seeds (O.init_params is deterministic numpy)."""
import os

import numpy as np
import pytest
import torch

from oracle import nerfies_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


def ref(name):
  return dict(np.load(os.path.join(HERE, 'golden', f'ref_{name}.npz'), allow_pickle=False))


T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)


def close(a, b, tol=1e-10, msg=''):
  np.testing.assert_allclose(a.detach().numpy() if torch.is_tensor(a) else a, b, rtol=tol, atol=tol, err_msg=msg)


def test_rigid_body():
  r = ref('rigid_body')
  close(O.exp_se3(T(r['screw']), T(r['theta'])), r['exp_se3'])
  close(O.skew(T(r['skew_in'])), r['skew'])
  close(O.exp_so3(T(r['screw'][0, :3]), T(r['theta'][0])), r['exp_so3'])


def test_sample_along_rays():
  r = ref('model_utils')
  for s in (0, 1):
    for l in (0, 1):
      z, pts = O.sample_along_rays(T(r['origins']), T(r['directions']), 12, float(r['near']), float(r['far']), bool(s), bool(l),
                                   T(r['t_rand']))
      close(z, r[f'sample_z_s{s}_l{l}']); close(pts, r[f'sample_pts_s{s}_l{l}'])


def test_volumetric_rendering_and_depth():
  r = ref('model_utils')
  for w in (0, 1):
    for i in (0, 1):
      out = O.volumetric_rendering(T(r['vr_rgb']), T(r['vr_sigma']), T(r['vr_z']), T(r['directions']), bool(w), bool(i))
      for k in ('rgb', 'depth', 'med_depth', 'acc', 'weights'):
        close(out[k], r[f'vr_w{w}_i{i}_{k}'], msg=f'w{w} i{i} {k}')
  w = T(r['pdf_weights'])
  assert (O.compute_depth_index(w).numpy() == r['depth_index']).all()
  close(O.compute_depth_map(w, T(r['vr_z'])), r['depth_map'])
  close(O.compute_opaqueness_mask(w), r['opaqueness_mask'])


def test_piecewise_constant_pdf_and_sample_pdf():
  r = ref('model_utils')
  z = T(r['vr_z']); w = T(r['pdf_weights'])
  z_mid = .5 * (z[..., 1:] + z[..., :-1])
  for s in (0, 1):
    zs = O.piecewise_constant_pdf(z_mid, w[..., 1:-1], 9, bool(s), T(r['u']))
    close(zs, r[f'pdf_z_s{s}'])
    zf, pf = O.sample_pdf(z_mid, w[..., 1:-1], T(r['origins']), T(r['directions']), z, 9, bool(s), T(r['u']))
    close(zf, r[f'sample_pdf_z_s{s}']); close(pf, r[f'sample_pdf_pts_s{s}'])


def test_encoders():
  r = ref('modules')
  x = T(r['x'])
  for F in (0, 4, 8):
    close(O.sinusoidal_encode(x, F), r[f'posenc_F{F}'])
  for a in (0.0, 2.5, 8.0):
    close(O.annealed_sinusoidal_encode(x, 8, a), r[f'annealed_a{a}'])
  close(O.cosine_easing_window(8, 3.25, torch.float64), r['window_a3.25'])


def test_nerf_mlp():
  r = ref('modules')
  spec = O.ModelSpec(use_camera_metadata=True)
  p = O.init_params(spec, seed=4, trained_like=True)['nerf_mlps_coarse']
  rgb, alpha = O.nerf_mlp(p, T(r['mlp_in']), None, T(r['mlp_cond']), spec)
  close(rgb, r['mlp_rgb'], 1e-9); close(alpha, r['mlp_alpha'], 1e-9)


def test_se3_field_and_jacobian():
  r = ref('se3_field')
  spec = O.ModelSpec(use_warp=True, num_warp_freqs=6, num_warp_features=8, num_warp_embeddings=4)
  wp = O.init_params(spec, seed=6, trained_like=True)['warp_field']
  out = O.se3_field(wp, T(r['points']), torch.tensor(r['ids']), float(r['alpha']), 6, return_jacobian=True)
  close(out['warped_points'], r['warped'], 1e-9)
  close(out['jacobian'], r['jacobian_fd'], 2e-6)   # reference side: central differences of ITS warp (jax.jacfwd in the original)


def test_translation_field_and_jacobian():
  """warping.TranslationField (warping.py:62-199), run by the reference on the oracle's parameter tree."""
  r = ref('translation_field')
  spec = O.ModelSpec(use_warp=True, warp_field_type='translation', num_warp_freqs=5, num_warp_features=8, num_warp_embeddings=4)
  wp = O.init_params(spec, seed=16, trained_like=True)['warp_field']
  assert set(wp) == {'metadata_encoder', 'mlp'} and set(wp['mlp']) == {f'hidden_{i}' for i in range(6)} | {'logit'}
  out = O.se3_field(wp, T(r['points']), torch.tensor(r['ids']), float(r['alpha']), 5, return_jacobian=True)
  close(out['warped_points'], r['warped'], 1e-10)
  close(out['jacobian'], r['jacobian_fd'], 2e-6)
  assert (out['warped_points'] - T(r['points'])).abs().max() > 1e-3      # the field actually moves the points


@pytest.mark.parametrize('name', ['nowarp', 'camera', 'warp'])
def test_nerf_model_apply_end_to_end(name):
  import sys
  sys.path.insert(0, os.path.join(HERE, 'golden'))
  import make_golden  # noqa: F401  (only for sys.path symmetry)
  cases = {
      'nowarp': (dict(num_coarse_samples=10, num_fine_samples=7, num_nerf_point_freqs=6, use_stratified_sampling=True), 0.0),
      'camera': (dict(num_coarse_samples=8, num_fine_samples=8, num_nerf_point_freqs=4, use_stratified_sampling=False,
                      use_camera_metadata=True), 0.0),
      'warp': (dict(num_coarse_samples=8, num_fine_samples=6, num_nerf_point_freqs=4, use_stratified_sampling=True, use_warp=True,
                    num_warp_freqs=5, num_warp_features=8, use_camera_metadata=True), 3.25),
  }
  kw, alpha = cases[name]
  r = ref('nerf_' + name)
  spec = O.ModelSpec(**kw)
  seed = int(r['seed'])
  params = O.init_params(spec, seed=seed, trained_like=True)
  batch = O.synthetic_batch(3, seed=seed + 1)
  ret = O.nerf_model_apply(params, spec, batch, alpha, return_points=spec.use_warp, return_warp_jacobian=spec.use_warp,
                           t_rand=T(r['t_rand']), u=T(r['u']))
  for lv in ('coarse', 'fine'):
    for k in ('rgb', 'depth', 'med_depth', 'acc', 'weights'):
      close(ret[lv][k], r[f'{lv}/{k}'], 1e-8, msg=f'{name} {lv}/{k}')
    if spec.use_warp:
      close(ret[lv]['points'], r[f'{lv}/points'], 1e-10); close(ret[lv]['warped_points'], r[f'{lv}/warped_points'], 1e-9)
      close(ret[lv]['warp_jacobian'], r[f'{lv}/warp_jacobian'], 2e-6)
  if spec.use_warp:
    el, res = O.compute_elastic_loss(ret['coarse']['warp_jacobian'])
    close(el, r['coarse/elastic_loss'], 2e-6); close(res, r['coarse/elastic_residual'], 2e-6)


NERF_CASES_R4 = {   # tests/golden/make_reference_vectors.py::nerf_model_r4 -- no condition at all: no bottleneck layer (modules.py:149-164)
    'nocond': (dict(num_coarse_samples=9, num_fine_samples=7, num_nerf_point_freqs=5, use_stratified_sampling=True, use_viewdirs=False), 0.0),
    'nocond_warp': (dict(num_coarse_samples=8, num_fine_samples=6, num_nerf_point_freqs=4, use_stratified_sampling=False, use_viewdirs=False,
                         use_warp=True, num_warp_freqs=5, num_warp_features=8), 2.75),
    # trunks shallower than 8 layers (modules.MLP, modules.py:41-62): the skip at layer 4 still inside / never reached
    'depth6': (dict(num_coarse_samples=8, num_fine_samples=6, num_nerf_point_freqs=4, use_stratified_sampling=True, nerf_trunk_depth=6,
                    use_camera_metadata=True), 0.0),
    'depth3': (dict(num_coarse_samples=8, num_fine_samples=6, num_nerf_point_freqs=4, use_stratified_sampling=True, nerf_trunk_depth=3), 0.0),
}


@pytest.mark.parametrize('name', sorted(NERF_CASES_R4))
def test_nerf_model_without_any_condition_and_shallow_trunks(name):
  kw, alpha = NERF_CASES_R4[name]
  r = ref('nerf_' + name)
  spec = O.ModelSpec(**kw)
  seed = int(r['seed'])
  params = O.init_params(spec, seed=seed, trained_like=True)
  assert ('bottleneck' in params['nerf_mlps_coarse']) == (not name.startswith('nocond'))
  assert len(params['nerf_mlps_coarse']['MLP_0']) == spec.nerf_trunk_depth
  batch = O.synthetic_batch(3, seed=seed + 1)
  ret = O.nerf_model_apply(params, spec, batch, alpha, return_points=spec.use_warp, return_warp_jacobian=spec.use_warp,
                           t_rand=T(r['t_rand']), u=T(r['u']))
  for lv in ('coarse', 'fine'):
    for k in ('rgb', 'depth', 'med_depth', 'acc', 'weights'):
      close(ret[lv][k], r[f'{lv}/{k}'], 1e-8, msg=f'{name} {lv}/{k}')
    if spec.use_warp:
      close(ret[lv]['warped_points'], r[f'{lv}/warped_points'], 1e-9)
      close(ret[lv]['warp_jacobian'], r[f'{lv}/warp_jacobian'], 2e-6)


NERF_CASES_R6 = {   # tests/golden/make_reference_vectors.py::nerf_model_r6 -- nerf_skips at another layer, warp_kwargs trunk shapes
    'skip5': (dict(num_coarse_samples=8, num_fine_samples=6, num_nerf_point_freqs=4, use_stratified_sampling=True, nerf_skips=(5,),
                   use_camera_metadata=True), 0.0),
    'skip2_depth6': (dict(num_coarse_samples=8, num_fine_samples=6, num_nerf_point_freqs=4, use_stratified_sampling=False, nerf_skips=(2,),
                          nerf_trunk_depth=6), 0.0),
    'skip1_warp': (dict(num_coarse_samples=8, num_fine_samples=6, num_nerf_point_freqs=4, use_stratified_sampling=True, nerf_skips=(1,),
                        use_warp=True, num_warp_freqs=5, num_warp_features=8), 2.5),
    'warp_trunk5x96': (dict(num_coarse_samples=8, num_fine_samples=6, num_nerf_point_freqs=4, use_stratified_sampling=True, use_warp=True,
                            num_warp_freqs=5, num_warp_features=8, warp_trunk_depth=5, warp_trunk_width=96, use_camera_metadata=True), 3.25),
    'warp_trunk3x64': (dict(num_coarse_samples=8, num_fine_samples=6, num_nerf_point_freqs=4, use_stratified_sampling=False, use_warp=True,
                            num_warp_freqs=4, num_warp_features=8, warp_trunk_depth=3, warp_trunk_width=64), 1.5),
    'translation_trunk4x80': (dict(num_coarse_samples=8, num_fine_samples=6, num_nerf_point_freqs=4, use_stratified_sampling=True,
                                   use_warp=True, warp_field_type='translation', num_warp_freqs=5, num_warp_features=8,
                                   warp_trunk_depth=4, warp_trunk_width=80), 2.25),
}


@pytest.mark.parametrize('name', sorted(NERF_CASES_R6))
def test_nerf_model_with_moved_skips_and_warp_kwargs(name):
  """modules.MLP skips (modules.py:47-48) at a layer other than 4; SE3Field / TranslationField trunks of the depth and width
  ModelConfig.warp_kwargs give them (configs.py:105, models.py:165-184, warping.py:225-226, 90-91): the reference was run with them."""
  kw, alpha = NERF_CASES_R6[name]
  r = ref('nerf_' + name)
  spec = O.ModelSpec(**kw)
  seed = int(r['seed'])
  params = O.init_params(spec, seed=seed, trained_like=True)
  if spec.use_warp:
    trunk = params['warp_field']['mlp' if spec.warp_field_type == 'translation' else 'trunk']
    assert sum(k.startswith('hidden_') for k in trunk) == spec.warp_trunk_depth
    assert trunk['hidden_0']['kernel'].shape[1] == spec.warp_trunk_width
  batch = O.synthetic_batch(3, seed=seed + 1)
  ret = O.nerf_model_apply(params, spec, batch, alpha, return_points=spec.use_warp, return_warp_jacobian=spec.use_warp,
                           t_rand=T(r['t_rand']), u=T(r['u']))
  for lv in ('coarse', 'fine'):
    for k in ('rgb', 'depth', 'med_depth', 'acc', 'weights'):
      close(ret[lv][k], r[f'{lv}/{k}'], 1e-8, msg=f'{name} {lv}/{k}')
    if spec.use_warp:
      close(ret[lv]['warped_points'], r[f'{lv}/warped_points'], 1e-9)
      close(ret[lv]['warp_jacobian'], r[f'{lv}/warp_jacobian'], 2e-6)


def test_oracle_at_the_full_config_a_batch_against_the_reference_run():
  """tests/golden/make_reference_vectors.py::nerf_model_full_batches: NerfModel.apply by the unmodified reference on the FULL 1024-ray batch
  of BASELINE.json configs[1] (64 + 128 samples, F_p = 8, stratified); the oracle on the same rays, parameters and re-drawn uniforms.
  (The 768- and 512-ray warp configurations are compared on the GPU only: tests/test_gpu_reference_onehop.py.)"""
  r = ref('nerf_cfgA_full')
  spec = O.ModelSpec(num_coarse_samples=64, num_fine_samples=128, num_nerf_point_freqs=8, use_stratified_sampling=True)
  seed, B = int(r['seed']), int(r['num_rays'])
  assert B == 1024
  params = O.init_params(spec, seed=seed, trained_like=True)
  batch = O.synthetic_batch(B, seed=seed + 1)
  rng = np.random.default_rng(seed + 2)
  t_rand = T(rng.uniform(0, 1, (B, spec.num_coarse_samples)).astype(np.float32).astype(np.float64))
  u = T(rng.uniform(0, 1, (B, spec.num_fine_samples)).astype(np.float32).astype(np.float64))
  with torch.no_grad():
    ret = O.nerf_model_apply(params, spec, batch, 0.0, t_rand=t_rand, u=u)
  for lv in ('coarse', 'fine'):
    for k in ('rgb', 'depth', 'acc'):
      close(ret[lv][k], r[f'{lv}/{k}'].astype(np.float64), 1e-6, msg=f'cfgA full {lv}/{k}')   # float32 storage of the fixture


def test_losses_psnr_elastic():
  r = ref('losses_schedules')
  sq = T(r['sq'])
  close(O.general_loss_with_squared_residual(sq, -2.0, 0.03), r['gl_m2_c03'], 1e-12)
  close(O.general_loss_with_squared_residual(sq, -2.0, 0.001), r['gl_m2_c001'], 1e-12)
  close(O.general_loss_with_squared_residual(sq, 1.0, 1.0), r['gl_1_c1'], 1e-12)
  close(O.compute_psnr(T([0.5, 0.01, 1e-4])), r['psnr'], 1e-12)
  el, res = O.compute_elastic_loss(T(r['el_J']))
  close(el, r['el_loss'], 1e-10); close(res, r['el_residual'], 1e-10)


def test_general_loss_every_branch():
  """utils.general_loss_with_squared_residual (utils.py:304-329): alpha = -inf, 0, 2, +inf and generic alphas, against the
  reference's own function (tests/golden/make_reference_vectors.py general_loss_branches)."""
  import math
  r = ref('general_loss_branches')
  sq = T(r['sq'])
  for name, alpha in (('neginf', -math.inf), ('m2', -2.0), ('zero', 0.0), ('one', 1.0), ('two', 2.0), ('posinf', math.inf)):
    for cname, scale in (('c03', 0.03), ('c1', 1.0)):
      got = O.general_loss_with_squared_residual(sq, alpha, scale).numpy()
      np.testing.assert_allclose(got, r[f'{name}_{cname}'], rtol=1e-12, atol=1e-300)


def test_background_loss():
  """training.compute_background_loss (training.py:117-135): the reference run on the oracle's warp parameters with the
  ids and the noise it draws supplied to it."""
  r = ref('background_loss')
  spec = O.ModelSpec(use_warp=True, num_warp_freqs=6, num_warp_features=8, num_warp_embeddings=4)
  params = O.init_params(spec, seed=21, trained_like=True)
  loss = O.compute_background_loss(params, spec, T(r['points']), torch.tensor(r['ids']), T(r['noise']) * float(r['noise_std']),
                                   float(r['alpha']))
  close(loss, r['loss'], 1e-12)
  assert float(loss.mean()) > 0


def test_render_image_tiling_matches_reference():
  """nerfies_amd.evaluation.render_image (CPU tensors, single process) against the reference's evaluation.render_image
  driven with the same per-ray stand-in model: same pixel order, same output dictionary, same shapes."""
  import types
  from nerfies_amd import evaluation
  r = ref('render_image')
  rays = {'origins': T(r['origins']), 'directions': T(r['directions']), 'metadata': {'warp': torch.tensor(r['warp'])}}

  def model_fn(key_0, key_1, params, rr, warp_extra):
    rgb = torch.tanh(rr['origins'] + 0.5 * rr['directions']) + 0.1 * rr['metadata']['warp'].double()
    return {'fine': {'rgb': rgb, 'depth': (rr['origins'] * rr['directions']).sum(-1), 'acc': rr['origins'][..., 0].abs()}}
  state = types.SimpleNamespace(optimizer=types.SimpleNamespace(target=None), warp_extra={})
  for chunk in (int(r['chunk']), 7, 100):          # the result does not depend on the chunk size
    out = evaluation.render_image(state, rays, model_fn, int(r['devices']), 0, chunk=chunk)
    assert set(out) == {'rgb', 'depth', 'acc'}
    for k in out:
      assert tuple(out[k].shape) == r['out/' + k].shape
      close(out[k], r['out/' + k], 1e-12)


def test_dataset_reader_matches_reference_datasource(tmp_path):
  """nerfies_amd.datasets.NerfiesDataSource against the reference's own NerfiesDataSource (run under the shim) on the
  same synthetic capture: ids, metadata vocabularies and table rows, per-item camera (rescaled + scene-normalised),
  decoded rgb, background points; and the oracle's camera_to_rays against datasets.core.camera_to_rays."""
  from nerfies_amd import datasets
  from oracle import camera_oracle as CO
  r = ref('dataset_items')
  d = str(tmp_path / 'cap')
  ids = datasets.write_synthetic_scene(d, num_frames=5, size=(16, 12), image_scale=2, seed=3)
  src = datasets.NerfiesDataSource(d, image_scale=2, use_appearance_id=True, use_camera_id=True, use_warp_id=True, random_seed=5)
  assert src.train_ids == list(r['train_ids']) and src.val_ids == list(r['val_ids'])
  assert src.appearance_ids == tuple(r['appearance_ids']) and src.camera_ids == tuple(r['camera_ids'])
  assert src.warp_ids == tuple(r['warp_ids']) and (src.near, src.far) == (float(r['near']), float(r['far']))
  np.testing.assert_array_equal(src.load_points(), r['points'])
  for i in ids:
    item = src.get_item(i)
    for k, v in item['camera'].get_parameters().items():
      np.testing.assert_allclose(np.asarray(v, np.float64), r[f'{i}/camera/{k}'].astype(np.float64), rtol=1e-6, atol=1e-7, err_msg=f'{i} {k}')
    np.testing.assert_array_equal(item['rgb'], r[f'{i}/rgb'])
    assert [item['metadata'][k] for k in ('appearance', 'camera', 'warp')] == list(r[f'{i}/metadata'])
  cam = src.load_camera(ids[2])
  ocam = CO.make_camera(cam.orientation, cam.position, cam.focal_length, cam.principal_point, [int(v) for v in cam.image_size],
                        cam.skew, cam.pixel_aspect_ratio, cam.radial_distortion, cam.tangential_distortion)
  rays = CO.camera_to_rays(ocam)
  np.testing.assert_allclose(rays['origins'], r['rays/origins'], atol=1e-7)
  np.testing.assert_allclose(rays['directions'], r['rays/directions'], atol=2e-6)      # the reference computes these in float32
  np.testing.assert_array_equal(rays['pixels'], r['rays/pixels'])


# ---- camera geometry and schedules ----
def _oracle_camera(r, tag, focal=None, pp=None, size=(320, 240), skew=None, par=None):
  from oracle import camera_oracle as CO
  f, cx, cy, sk, pa = r['intrinsics']
  return CO.make_camera(r[f'{tag}/orientation'], r[f'{tag}/position'], f if focal is None else focal,
                        [cx, cy] if pp is None else pp, size, sk if skew is None else skew, pa if par is None else par,
                        r[f'{tag}/radial'], r[f'{tag}/tangential'])


@pytest.mark.parametrize('tag', ['pinhole', 'distorted'])
def test_camera_oracle_matches_reference_camera(tag):
  from oracle import camera_oracle as CO
  r = ref('camera')
  cam = _oracle_camera(r, tag)
  np.testing.assert_allclose(CO.pixels_to_rays(cam, r[f'{tag}/pixels']), r[f'{tag}/rays'], rtol=0, atol=1e-13)
  np.testing.assert_allclose(CO.project(cam, r[f'{tag}/points']), r[f'{tag}/project'], rtol=0, atol=1e-10)
  # undistort really inverts distort: projecting points on the rays returns the pixels
  np.testing.assert_allclose(CO.project(cam, r[f'{tag}/points']), r[f'{tag}/pixels'], rtol=0, atol=1e-9)
  small = _oracle_camera(r, tag, focal=20.0, pp=[3.5, 2.5], size=(7, 5), skew=0.0, par=1.0)
  np.testing.assert_allclose(CO.pixel_centers(small), r[f'{tag}/centers_7x5'], rtol=0, atol=0)
  np.testing.assert_allclose(CO.pixels_to_rays(small, CO.pixel_centers(small)), r[f'{tag}/centers_rays_7x5'], rtol=0,
                             atol=1e-13)
  rays = CO.camera_to_rays(small)
  assert rays['origins'].shape == (5, 7, 3) and rays['directions'].dtype == np.float32


def test_schedules_match_reference():
  from nerfies_amd import schedules as S
  r = ref('losses_schedules')
  steps = [int(s) for s in r['sched_steps']]
  defs = {
      'constant': ('constant', 0.3),
      'linear': ('linear', 0.0, 8.0, 80000),
      'exponential': ('exponential', 1e-3, 1e-4, 250000),
      'cosine_easing': ('cosine_easing', 0.01, 1e-8, 5000),
      'piecewise': ('piecewise', [(500, ('constant', 0.01)), (2000, ('cosine_easing', 0.01, 1e-5, 2000)),
                                  (1, ('constant', 1e-5))]),
      'delayed': ('delayed', ('exponential', 1e-3, 1e-4, 250000), 2500, 0.01),
      'step': ('step', 1e-3, 1000, 0.5, 3),
      'dict_linear': {'type': 'linear', 'initial_value': 1.0, 'final_value': 0.25, 'num_steps': 1000},
  }
  for name, dfn in defs.items():
    sch = S.from_config(dfn)
    got = np.array([sch(s) for s in steps])
    # the reference rounds its flat segments to float32 (jnp.full_like(..., dtype=float32)): 6e-8 relative
    np.testing.assert_allclose(got, r['sched_' + name], rtol=1e-7, atol=0, err_msg=name)
  assert S.from_config(S.ConstantSchedule(2.0))(7) == 2.0
  with pytest.raises(ValueError):
    S.ExponentialSchedule(1e-4, 1e-3, 10)
  with pytest.raises(ValueError):
    S.from_config(3.0)
  with pytest.raises(KeyError):
    S.from_tuple(('nope', 1))


# ---- round 2: the rest of NerfModel.apply's contract and train_step's own loss assembly ----
NERF_CASES_R2 = {
    'alpha_cond': (dict(num_coarse_samples=8, num_fine_samples=6, num_nerf_point_freqs=4, use_stratified_sampling=True,
                        use_appearance_metadata=True, use_alpha_condition=True, use_camera_metadata=True), 0.0),
    'encoded': (dict(num_coarse_samples=8, num_fine_samples=6, num_nerf_point_freqs=4, use_stratified_sampling=False, use_warp=True,
                     num_warp_freqs=5, num_warp_features=8, use_camera_metadata=True, use_appearance_metadata=True,
                     use_alpha_condition=True), 3.25),
    'time': (dict(num_coarse_samples=8, num_fine_samples=6, num_nerf_point_freqs=4, use_stratified_sampling=True, use_warp=True,
                  num_warp_freqs=5, num_warp_features=8, warp_metadata_encoder_type='time'), 3.25),
}


@pytest.mark.parametrize('name', sorted(NERF_CASES_R2))
def test_nerf_model_apply_alpha_condition_and_encoded_metadata(name):
  """NerfModel.apply with use_alpha_condition (modules.py:152-157 and the models.py:206 quirk) and with
  metadata_encoded=True (models.py:198-199, 210-211, 251; warping.py:378-381), as run by the reference itself."""
  kw, alpha = NERF_CASES_R2[name]
  r = ref('nerf_' + name)
  spec = O.ModelSpec(**kw)
  seed = int(r['seed'])
  params = O.init_params(spec, seed=seed, trained_like=True)
  batch = O.synthetic_batch(3, seed=seed + 1)
  encoded = name == 'encoded'
  if encoded:
    ids = {k: v[:, 0] for k, v in batch['metadata'].items()}
    batch['metadata'] = {k: T(r['codes/' + k]) for k in ('warp', 'appearance', 'camera')}
    close(params['appearance_encoder']['embed']['embedding'][ids['appearance']], r['codes/appearance'], 1e-15)
  ret = O.nerf_model_apply(params, spec, batch, alpha, metadata_encoded=encoded, return_points=spec.use_warp,
                           return_warp_jacobian=spec.use_warp, t_rand=T(r['t_rand']), u=T(r['u']), time_alpha=float(r['time_alpha']))
  if name == 'time':   # modules.TimeEncoder (modules.py:297-322) through SE3Field.encode_metadata (warping.py:311-313)
    assert set(params['warp_field']['metadata_encoder']) == {'mlp'}
    other = O.nerf_model_apply(params, spec, batch, alpha, t_rand=T(r['t_rand']), u=T(r['u']), time_alpha=2.0)
    assert (other['fine']['rgb'] - ret['fine']['rgb']).abs().max() > 1e-6    # the annealing window of the time posenc matters
  for lv in ('coarse', 'fine'):
    for k in ('rgb', 'depth', 'med_depth', 'acc', 'weights'):
      close(ret[lv][k], r[f'{lv}/{k}'], 1e-8, msg=f'{name} {lv}/{k}')
    if spec.use_warp:
      close(ret[lv]['warped_points'], r[f'{lv}/warped_points'], 1e-9)
      close(ret[lv]['warp_jacobian'], r[f'{lv}/warp_jacobian'], 2e-6)
  if encoded:   # ... and the encoded evaluation equals the id-driven one
    batch2 = O.synthetic_batch(3, seed=seed + 1)
    ret2 = O.nerf_model_apply(params, spec, batch2, alpha, t_rand=T(r['t_rand']), u=T(r['u']))
    close(ret2['fine']['rgb'], r['fine/rgb'], 1e-8)
  # the appearance code must matter: without the alpha condition it is dead (models.py:204-208)
  if name == 'time':
    return
  p2 = O.tree_map(lambda t: t.clone(), params)
  p2['appearance_encoder']['embed']['embedding'] += 0.3
  if not encoded:
    ret3 = O.nerf_model_apply(p2, spec, batch, alpha, t_rand=T(r['t_rand']), u=T(r['u']))
    assert (ret3['fine']['rgb'] - ret['fine']['rgb']).abs().max() > 1e-4


TRAIN_CASES = {'log_svals_weight': ('log_svals', 'weight', True), 'svals_median': ('svals', 'median', False),
               'jtj_weight': ('jtj', 'weight', False), 'div_weight': ('div', 'weight', True),
               'det_median': ('det', 'median', False), 'log_det_weight': ('log_det', 'weight', False)}


@pytest.mark.parametrize('name', sorted(TRAIN_CASES))
def test_train_step_loss_assembly_matches_reference(name):
  """The forward half of the reference's own training.train_step (training.py:168-262) -- run under the shim with
  jax.value_and_grad replaced by a plain evaluation -- against oracle.loss_fn: rgb losses, psnr, every elastic_loss_type
  under both reduce methods, warp_reg, background, Jacobian metrics, per-level totals."""
  ltype, method, wreg = TRAIN_CASES[name]
  r = ref('train_step_stats')
  spec = O.ModelSpec(num_coarse_samples=8, num_fine_samples=6, num_nerf_point_freqs=4, use_stratified_sampling=True, use_warp=True,
                     num_warp_freqs=5, num_warp_features=8, use_camera_metadata=True)
  params = O.init_params(spec, seed=41, trained_like=True)
  batch = O.synthetic_batch(3, seed=42)
  bg = {'points': T(r['bg_points']), 'warp_ids': torch.tensor(r['bg_ids']), 'noise': T(r['bg_noise']) * 0.001}
  total, stats, _ = O.loss_fn(params, spec, batch, float(r['alpha']), t_rand=T(r['t_rand']), u=T(r['u']), use_elastic_loss=True,
                              elastic_loss_weight=float(r['elastic_loss_weight']), elastic_reduce_method=method,
                              elastic_loss_type=ltype, use_background_loss=True,
                              background_loss_weight=float(r['background_loss_weight']), background=bg, use_warp_reg_loss=wreg,
                              warp_reg_loss_weight=float(r['warp_reg_loss_weight']))
  seen = 0
  for lv in ('coarse', 'fine'):
    keys = [k[len(f'{name}/{lv}/'):] for k in r if k.startswith(f'{name}/{lv}/')]
    assert set(keys) == set(stats[lv]), (sorted(keys), sorted(stats[lv]))
    for k in keys:
      # the reference Jacobians are central differences of its warp (~1e-7); everything derived from them inherits that
      tol = 1e-9 if k in ('loss/rgb', 'metric/psnr') or (lv == 'fine' and 'warp_reg' not in k and k != 'loss/total') else 5e-6
      close(stats[lv][k], r[f'{name}/{lv}/{k}'], tol, msg=f'{name} {lv} {k}')
      seen += 1
  close(stats['background_loss'], r[f'{name}/background_loss'], 1e-10)
  assert seen >= 11 and ('loss/warp_reg' in stats['fine']) == wreg
  want_total = sum(float(r[f'{name}/{lv}/loss/total']) for lv in ('coarse', 'fine')) + float(r['background_loss_weight']) * float(r[f'{name}/background_loss'])
  assert abs(total.item() - want_total) < 1e-6


def test_elastic_loss_types_and_noise_regularize():
  r = ref('elastic_types_noise')
  J = T(r['J'])
  close(O.jacobian_to_div(J), r['div'], 1e-12); close(O.jacobian_to_curl(J), r['curl'], 1e-12)
  for t in ('log_svals', 'svals', 'jtj', 'div', 'det', 'log_det'):
    el, res = O.compute_elastic_loss(J, loss_type=t)
    close(el, r[f'{t}/loss'], 1e-10, msg=t); close(res, r[f'{t}/residual'], 1e-10, msg=t)
  with pytest.raises(NotImplementedError):
    O.compute_elastic_loss(J, loss_type='nr')
  raw, nz = T(r['raw']), T(r['normals'])
  close(torch.cat([raw[..., :3], O.noise_regularize(raw[..., 3:4], 0.4, True, nz)], -1), r['noised_strat'], 1e-15)
  close(torch.cat([raw[..., :3], O.noise_regularize(raw[..., 3:4], 0.4, False, nz)], -1), r['noised_det'], 0)
  close(torch.cat([raw[..., :3], O.noise_regularize(raw[..., 3:4], None, True, nz)], -1), r['noised_none'], 0)


@pytest.mark.parametrize('name', ['nowarp', 'warp_bg'])
def test_oracle_gradient_against_the_reference_side_directional_derivative(name):
  """The one reference-side GRADIENT evidence the NumPy shim allows (jax.value_and_grad cannot run): central differences, in
  float64, of the reference's own `_loss_fn` closure (training.py:229-262) along 8 seeded parameter directions, with
  lax.stop_gradient replayed from the base evaluation (model_utils.py:187, training.py:181) so that it means what it means under
  autodiff (tests/golden/make_reference_vectors.py::loss_directional).  The oracle's torch.autograd gradient -- what every GPU
  gradient test is measured against -- must reproduce <grad, v> for every direction."""
  import sys
  sys.path.insert(0, HERE)
  import helpers as H
  r = ref('loss_directional_' + name)
  case = H.LOSS_DIR_CASES[name]
  spec = O.ModelSpec(**case['spec'])
  seed, B = int(r['seed']), int(r['num_rays'])
  params = O.init_params(spec, seed=seed, trained_like=True)
  batch = O.synthetic_batch(B, seed=seed + 1)
  kw = dict(warp_alpha=float(r['alpha']), t_rand=T(r['t_rand']), u=T(r['u']))
  if case['bg']:
    kw.update(use_background_loss=True, background_loss_weight=float(r['background_loss_weight']),
              background={'points': T(r['bg_points']), 'warp_ids': torch.tensor(r['bg_ids']), 'noise': T(r['bg_noise']) * 0.001})
  loss, _, grads, _ = O.loss_and_grad(params, spec, batch, **kw)
  close(loss, r['loss'], 1e-9)
  dirs = H.loss_directions(params, int(r['dir_seed']), len(r['directional']))
  got = np.array([H.tree_dot(grads, d) for d in dirs])
  want = r['directional']
  # the reference side is a central difference (eps per case, see the generator): truncation + cancellation ~1e-6 of |f'|
  np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-7, err_msg=name)
  assert np.abs(want).min() > 1e-4      # every direction has a real slope: not a comparison of zeros

