"""Split-bf16 ("bf16x3", NRF_FLAG_BF16X3) inference chains against the REFERENCE's own outputs, in one hop.

The mode evaluates the NeRF MLPs (modules.py:26-62, 95-169) with every float32 operand as a bf16 pair hi + lo and a product as
hi.hi + lo.hi + hi.lo on the bf16 matrix pipe (csrc/mlp_bf16x3.hip): float32-EMULATING, not bit-comparable with the float32 chains.
Since the SE3 trunk joined the mode (csrc/warp_bf16x3.hip) the warp cases run it in split-bf16 as well; bf16='x3mlp' keeps it float32.
What is held here: the arrays tests/golden/ref_nerf_*.npz (NerfModel.apply by the unmodified reference, models.py:289-375) to the
same tolerances as the float32 path's one-hop tests, rendered colour at the BASELINE shapes to 1e-5 (configuration A, no warp:
the gate VERDICT r5 item 4 names), and the distance to the library's own float32 path printed next to it."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import helpers as H  # noqa: E402
from nerfies_amd import lib as L  # noqa: E402
from oracle import nerfies_oracle as O  # noqa: E402
from test_gpu_reference_onehop import BASELINE_CASES, CASES, FULL_BATCH, _np, _ref  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda'

# the cases the bf16 chains can lay out (skip at layer 4); the moved-skip cases are float32-only (NRF_E_UNSUPPORTED, below)
SMALL = ['nowarp', 'camera', 'warp', 'nocond', 'nocond_warp', 'depth6', 'depth3', 'skip2_depth6', 'warp_trunk5x96', 'warp_trunk3x64',
         'translation_trunk4x80']


@pytest.mark.parametrize('name', SMALL)
def test_x3_against_the_reference_run(name):
  kw, alpha = CASES[name]
  r = _ref('nerf_' + name)
  spec = O.ModelSpec(**kw)
  seed = int(r['seed'])
  params = O.init_params(spec, seed=seed, trained_like=True)
  batch = O.synthetic_batch(3, seed=seed + 1)
  model, fp = H.gpu_model(spec, params, 3)
  rngs = {'coarse': torch.tensor(r['t_rand']).float().to(DEV), 'fine': torch.tensor(r['u']).float().to(DEV)}
  kwargs = dict(rngs=rngs, return_weights=True, return_points=spec.use_warp)
  out = model.apply({'params': fp}, H.gpu_batch(batch), {'alpha': alpha}, bf16='x3', **kwargs)
  f32 = model.apply({'params': fp}, H.gpu_batch(batch), {'alpha': alpha}, **kwargs)
  worst, vs32 = {}, 0.0
  for lv in ('coarse', 'fine'):
    for k in ('rgb', 'depth', 'acc', 'weights'):
      got, want = _np(out[lv][k]), r[f'{lv}/{k}']
      np.testing.assert_allclose(got, want, atol=1e-4, err_msg=f'{name} {lv}/{k}')
      worst[k] = max(worst.get(k, 0.0), float(np.abs(got - want).max()))
      vs32 = max(vs32, float(np.abs(got - _np(f32[lv][k])).max()))
    if spec.use_warp:   # the SE3 trunk runs in split-bf16 as well (csrc/warp_bf16x3.hip): the warped points against the reference's
      dp = float(np.abs(_np(out[lv]['warped_points']) - r[f'{lv}/warped_points']).max())
      worst['warped_points'] = max(worst.get('warped_points', 0.0), dp)
      np.testing.assert_allclose(_np(out[lv]['warped_points']), r[f'{lv}/warped_points'], atol=2e-5)
      if lv == 'coarse':   # same sample positions on both paths: the distance of the two trunks
        vs32 = max(vs32, float((out[lv]['warped_points'] - f32[lv]['warped_points']).abs().max()))
  print(f'bf16x3 one-hop {name}: max |hip - reference| ' + ', '.join(f'{k} {v:.2e}' for k, v in worst.items()) +
        f'; max |x3 - float32 path| {vs32:.2e}')


@pytest.mark.parametrize('name', sorted(BASELINE_CASES))
def test_x3_at_the_baseline_shapes_against_the_reference_run(name):
  """64 rays x (64 + 128) at F_p = 8; 16 x (128 + 128) with the F_w = 6 warp; 8 x (256 + 256) at F_p = 10 with the F_w = 8 warp."""
  kw, alpha = BASELINE_CASES[name]
  r = _ref('nerf_' + name)
  spec = O.ModelSpec(**kw)
  seed, B = int(r['seed']), int(r['num_rays'])
  params = O.init_params(spec, seed=seed, trained_like=True)
  batch = O.synthetic_batch(B, seed=seed + 1)
  model, fp = H.gpu_model(spec, params, B)
  rngs = {'coarse': torch.tensor(r['t_rand']).float().to(DEV), 'fine': torch.tensor(r['u']).float().to(DEV)}
  out = model.apply({'params': fp}, H.gpu_batch(batch), {'alpha': alpha}, rngs=rngs, return_weights=True, bf16='x3')
  f32 = model.apply({'params': fp}, H.gpu_batch(batch), {'alpha': alpha}, rngs=rngs, return_weights=True)
  tol = 1e-3 if spec.use_warp else 1e-5   # no warp: the 1e-5 gate of the mode; with it the float32 warp's own rounding meets the posenc band
  worst, w32, vs32 = {}, {}, {}
  for lv in ('coarse', 'fine'):
    for k in ('rgb', 'depth', 'acc'):
      got, want = _np(out[lv][k]), r[f'{lv}/{k}'].astype(np.float64)
      worst[k] = max(worst.get(k, 0.0), float(np.abs(got - want).max()))
      w32[k] = max(w32.get(k, 0.0), float(np.abs(_np(f32[lv][k]) - want).max()))
      vs32[k] = max(vs32.get(k, 0.0), float(np.abs(got - _np(f32[lv][k])).max()))
      np.testing.assert_allclose(got, want, atol=tol if k == 'rgb' or spec.use_warp else 1e-4, err_msg=f'{name} {lv}/{k}')
  print(f'bf16x3 one-hop {name} ({B} rays x {spec.num_coarse_samples}+{spec.num_fine_samples}): max |x3 - reference| ' +
        ', '.join(f'{k} {v:.2e}' for k, v in worst.items()) + ' (float32 path: ' + ', '.join(f'{k} {v:.2e}' for k, v in w32.items()) +
        '); max |x3 - float32 path| ' + ', '.join(f'{k} {v:.2e}' for k, v in vs32.items()))


def test_x3_at_the_full_config_a_batch_against_the_reference_run():
  """1024 rays x (64 + 128): BASELINE configs[1] at its full batch, rendered by the unmodified reference."""
  name = 'cfgA'
  kw, alpha = BASELINE_CASES[name]
  r = _ref('nerf_' + name + '_full')
  spec = O.ModelSpec(**kw)
  seed, B = int(r['seed']), int(r['num_rays'])
  assert B == FULL_BATCH[name]
  params = O.init_params(spec, seed=seed, trained_like=True)
  batch = O.synthetic_batch(B, seed=seed + 1)
  rng = np.random.default_rng(seed + 2)
  t_rand = rng.uniform(0, 1, (B, spec.num_coarse_samples)).astype(np.float32)
  u = rng.uniform(0, 1, (B, spec.num_fine_samples)).astype(np.float32)
  model, fp = H.gpu_model(spec, params, B)
  rngs = {'coarse': torch.tensor(t_rand).to(DEV), 'fine': torch.tensor(u).to(DEV)}
  out = model.apply({'params': fp}, H.gpu_batch(batch), {'alpha': alpha}, rngs=rngs, bf16='x3')
  worst = {}
  for lv in ('coarse', 'fine'):
    for k in ('rgb', 'depth', 'acc'):
      d = np.abs(_np(out[lv][k]) - r[f'{lv}/{k}'].astype(np.float64))
      worst[k] = max(worst.get(k, 0.0), float(d.max()))
      # as the float32 test: a stratified inverse-CDF draw within rounding of a bin edge may displace one fine sample of a few rays
      assert (d > 1e-4).sum() <= max(1, d.size // 500), (lv, k, float(d.max()), int((d > 1e-4).sum()))
      assert np.quantile(d, 0.99) <= 1e-5, (lv, k, float(np.quantile(d, 0.99)))
  print(f'bf16x3 one-hop cfgA at the full batch ({B} rays): max |hip - reference| ' + ', '.join(f'{k} {v:.2e}' for k, v in worst.items()))


def test_x3_alpha_condition_matches_the_float32_path():
  """use_alpha_condition (modules.py:152-157): the alpha head reads [bottleneck, appearance code] -- the kernel's ABN variant."""
  spec = O.ModelSpec(num_coarse_samples=12, num_fine_samples=10, num_nerf_point_freqs=6, use_stratified_sampling=False,
                     use_appearance_metadata=True, use_alpha_condition=True)
  params = O.init_params(spec, seed=5, trained_like=True)
  batch = O.synthetic_batch(37, seed=6)
  model, fp = H.gpu_model(spec, params, 37)
  out = model.apply({'params': fp}, H.gpu_batch(batch), {'alpha': 0.0}, bf16='x3')
  f32 = model.apply({'params': fp}, H.gpu_batch(batch), {'alpha': 0.0})
  for lv in ('coarse', 'fine'):
    for k in ('rgb', 'depth', 'acc'):
      np.testing.assert_allclose(_np(out[lv][k]), _np(f32[lv][k]), atol=2e-5, err_msg=f'{lv}/{k}')


def test_x3_with_the_float32_warp_trunk_keeps_the_warped_points():
  """bf16='x3mlp' = NRF_FLAG_BF16X3 | NRF_FLAG_WARP_F32: the SE3 trunk stays on the float32 kernels -- the coarse level's warped points are
  bit-identical to the float32 path's (the fine level's sample positions follow the coarse weights), colour within 1e-5 of it."""
  kw, alpha = CASES['warp']
  spec = O.ModelSpec(**kw)
  params = O.init_params(spec, seed=11, trained_like=True)
  batch = O.synthetic_batch(40, seed=12)
  model, fp = H.gpu_model(spec, params, 40)
  rngs = {'coarse': torch.rand(40, spec.num_coarse_samples, device=DEV), 'fine': torch.rand(40, spec.num_fine_samples, device=DEV)}
  a = model.apply({'params': fp}, H.gpu_batch(batch), {'alpha': alpha}, rngs=rngs, return_points=True, bf16='x3mlp')
  b = model.apply({'params': fp}, H.gpu_batch(batch), {'alpha': alpha}, rngs=rngs, return_points=True)
  c = model.apply({'params': fp}, H.gpu_batch(batch), {'alpha': alpha}, rngs=rngs, return_points=True, bf16='x3')
  assert torch.equal(a['coarse']['warped_points'], b['coarse']['warped_points'])
  assert not torch.equal(c['coarse']['warped_points'], b['coarse']['warped_points'])      # the split-bf16 trunk is another arithmetic
  np.testing.assert_allclose(_np(c['coarse']['warped_points']), _np(b['coarse']['warped_points']), atol=5e-5)
  for lv in ('coarse', 'fine'):
    np.testing.assert_allclose(_np(a[lv]['rgb']), _np(b[lv]['rgb']), atol=1e-5)
    np.testing.assert_allclose(_np(c[lv]['rgb']), _np(b[lv]['rgb']), atol=1e-4)


@pytest.mark.parametrize('kw,B,alpha,time_alpha', [
    (dict(num_nerf_point_freqs=8, num_warp_freqs=6, num_warp_features=5, use_camera_metadata=True, num_coarse_samples=32, num_fine_samples=32), 33, 4.5, 0.4),
    (dict(num_nerf_point_freqs=6, warp_field_type='translation', num_coarse_samples=24, num_fine_samples=24), 12, 2.0, 0.0)])
def test_x3_with_the_time_encoder_and_encoded_codes(kw, B, alpha, time_alpha):
  """The warp metadata as per-ray codes (modules.TimeEncoder's output, modules.py:297-322; or metadata_encoded=True, warping.py:378-381):
  the split-bf16 trunk gathers row `ray` of the code table; against the float32 path on the same rays."""
  spec = O.ModelSpec(use_warp=True, use_stratified_sampling=False, warp_metadata_encoder_type='time', **kw)
  p = O.init_params(spec, seed=17, trained_like=True)
  b = O.synthetic_batch(B, seed=18)
  model, fp = H.gpu_model(spec, p, B)
  gb = H.gpu_batch(b)
  extra = {'alpha': alpha, 'time_alpha': time_alpha}
  a = model.apply({'params': fp}, gb, extra, bf16='x3', return_points=True)
  f = model.apply({'params': fp}, gb, extra, return_points=True)
  np.testing.assert_allclose(_np(a['coarse']['warped_points']), _np(f['coarse']['warped_points']), atol=5e-5)
  for lv in ('coarse', 'fine'):
    np.testing.assert_allclose(_np(a[lv]['rgb']), _np(f[lv]['rgb']), atol=1e-4)
    np.testing.assert_allclose(_np(a[lv]['depth']), _np(f[lv]['depth']), atol=1e-4)
  if not spec.use_camera_metadata:   # metadata_encoded wants every metadata entry as codes: the case whose only entry is the time stamp
    codes = O.time_encode(p['warp_field']['metadata_encoder'], b['metadata']['time'].double(), spec.num_time_encoder_freqs, time_alpha)
    enc = dict(gb)
    enc['metadata'] = {'time': codes.float().to(DEV)}
    c = model.apply({'params': fp}, enc, extra, metadata_encoded=True, bf16='x3')
    np.testing.assert_allclose(_np(c['fine']['rgb']), _np(a['fine']['rgb']), atol=5e-5)


def test_x3_is_an_inference_mode_of_its_own():
  spec = O.ModelSpec(num_coarse_samples=8, num_fine_samples=8, num_nerf_point_freqs=4)
  params = O.init_params(spec, seed=1, trained_like=True)
  model, _ = H.gpu_model(spec, params, 4)
  import ctypes as C
  n = C.c_size_t(0)
  for flags in (L.NRF_FLAG_BF16X3 | L.NRF_FLAG_TRAIN, L.NRF_FLAG_BF16X3 | L.NRF_FLAG_BF16):
    assert model.lib.nrf_workspace_bytes(model.handle, 4, flags, C.byref(n)) != 0
  assert model.lib.nrf_workspace_bytes(model.handle, 4, L.NRF_FLAG_BF16X3, C.byref(n)) == 0 and n.value > 0
  # a moved skip is float32-only, as for NRF_FLAG_BF16
  spec5 = O.ModelSpec(num_coarse_samples=8, num_fine_samples=8, num_nerf_point_freqs=4, nerf_skips=(5,))
  model5, _ = H.gpu_model(spec5, O.init_params(spec5, seed=1, trained_like=True), 4)
  assert model5.lib.nrf_workspace_bytes(model5.handle, 4, L.NRF_FLAG_BF16X3, C.byref(n)) != 0
