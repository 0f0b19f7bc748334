"""The rest of NerfModel.apply's contract and of train_step's loss terms on the GPU:
use_alpha_condition, metadata_encoded, return_warp_jacobian / use_warp_jacobian, noise_std, use_warp_reg_loss, every
elastic_loss_type, the Jacobian metrics.  Forward results against the fp64 oracle AND against vectors produced by the
reference's own NerfModel.apply (tests/golden/ref_nerf_alpha_cond.npz, ref_nerf_encoded.npz); gradients leaf by leaf
against the oracle pinned to the HIP path's branch pattern (tests/test_gpu_pinned.py)."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from oracle import nerfies_oracle as O  # noqa: E402
import helpers as H  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _ref(name):
  return dict(np.load(os.path.join(HERE, 'golden', f'ref_{name}.npz')))


def _np(t):
  return t.detach().cpu().numpy()


# ---------------------------------------------------------------------------------------------
# use_alpha_condition (models.py:204-208, modules.py:152-157)
# ---------------------------------------------------------------------------------------------
ALPHA_KW = dict(use_appearance_metadata=True, use_alpha_condition=True, num_coarse_samples=32, num_fine_samples=32)


@pytest.mark.parametrize('kw,B', [(dict(use_camera_metadata=True), 11), (dict(use_stratified_sampling=True, num_appearance_features=5), 70),
                                  (dict(nerf_trunk_width=128, nerf_rgb_branch_width=64), 9),
                                  (dict(use_warp=True, num_nerf_point_freqs=6, use_camera_metadata=True), 9)])
def test_alpha_condition_forward_and_gradients(kw, B):
  spec = O.ModelSpec(**dict(ALPHA_KW, **kw))
  r = H.run_pinned(spec, B, 4.0, seed=31)
  H.assert_pinned(r, f'alpha condition {kw}')
  H.assert_forward(r, spec)
  e = r['errs']
  # the appearance table and the code rows of both heads carry gradient
  assert e['appearance_encoder/embed/embedding'][1] > 0
  assert e['nerf_mlps_fine/MLP_2/logit/kernel'][1] > 0 and e['nerf_mlps_coarse/MLP_1/hidden_0/kernel'][1] > 0
  from nerfies_amd import params as P
  tree = P.tree_from_flat(r['fp'].flat.cpu(), r['model'].layout)
  assert tuple(tree['nerf_mlps_fine']['MLP_2']['logit']['kernel'].shape) == (spec.nerf_trunk_width + spec.num_appearance_features, 1)
  # the bf16 mode with the alpha head on the bottleneck (round 4: the head's transposed row enters the dgrad chain at the
  # bottleneck GEMM, its weight gradient pairs with the bottleneck stash): every leaf keeps the float32 path's direction
  extra = dict(warp_extra={'alpha': 4.0}, rngs=r['rngs'])
  g32, s32 = r['model'].loss_and_grad(r['fp'], r['gb'], **extra)
  g32, s32 = g32.clone(), s32.clone()
  g16, s16 = r['model'].loss_and_grad(r['fp'], r['gb'], bf16='mlp', **extra)
  assert abs(s16[4].item() - s32[4].item()) < 1e-3 + 2e-2 * abs(s32[4].item())
  t32, t16 = P.tree_from_flat(g32.cpu(), r['model'].layout), P.tree_from_flat(g16.cpu(), r['model'].layout)
  worst = ('', 1.0)
  for path, a in O.tree_leaves_with_path(t32):
    if path.startswith('warp_field') or a.abs().max().item() < 1e-7:
      continue   # (the warp field sees the NeRF MLPs' rounding through the 2^(F_p - 1) posenc: tests/test_gpu_bf16_warp.py)
    c = torch.nn.functional.cosine_similarity(a.flatten().double(), H.leaf(t16, path).flatten().double(), dim=0).item()
    if c < worst[1]:
      worst = (path, c)
  print(f'[alpha condition {kw}, bf16 vs f32] loss {s16[4].item():.6f} / {s32[4].item():.6f}; worst leaf cosine {worst[1]:.4f} ({worst[0]})')
  assert worst[1] > 0.97, worst
  o16 = r['model'].apply({'params': r['fp']}, r['gb'], {'alpha': 4.0}, rngs=r['rngs'], bf16='mlp')
  o32 = r['model'].apply({'params': r['fp']}, r['gb'], {'alpha': 4.0}, rngs=r['rngs'])
  for lv in o32:
    assert (o16[lv]['rgb'] - o32[lv]['rgb']).abs().max().item() < 2e-2 and (o16[lv]['acc'] - o32[lv]['acc']).abs().max().item() < 2e-2


def test_alpha_condition_matches_the_reference_run():
  """GPU forward against the output of the reference's own NerfModel.apply (through the NumPy shim)."""
  r = _ref('nerf_alpha_cond')
  spec = O.ModelSpec(num_coarse_samples=8, num_fine_samples=6, num_nerf_point_freqs=4, use_stratified_sampling=True,
                     use_appearance_metadata=True, use_alpha_condition=True, use_camera_metadata=True)
  seed = int(r['seed'])
  params = O.init_params(spec, seed=seed, trained_like=True)
  batch = O.synthetic_batch(3, seed=seed + 1)
  model, fp = H.gpu_model(spec, params, 3)
  out = model.apply({'params': fp}, H.gpu_batch(batch), {'alpha': 0.0},
                    rngs={'coarse': torch.tensor(r['t_rand']).float().to(DEV), 'fine': torch.tensor(r['u']).float().to(DEV)},
                    return_weights=True)
  for lv in ('coarse', 'fine'):
    for k in ('rgb', 'depth', 'acc', 'weights'):
      np.testing.assert_allclose(_np(out[lv][k]), r[f'{lv}/{k}'], atol=1e-4, err_msg=f'{lv}/{k}')


def test_appearance_code_is_dead_without_alpha_condition():
  """models.py:204-208: with use_alpha_condition=False the appearance code reaches nothing (the rgb branch only gets it
  under the same flag) -- kept as in the reference."""
  spec = O.ModelSpec(use_appearance_metadata=True, use_alpha_condition=False, num_coarse_samples=16, num_fine_samples=16)
  p = O.init_params(spec, seed=2, trained_like=True)
  b = O.synthetic_batch(5, seed=3)
  model, fp = H.gpu_model(spec, p, 5)
  gb = H.gpu_batch(b)
  a = model.apply({'params': fp}, gb, {})['fine']['rgb'].clone()
  gb['metadata']['appearance'] = (gb['metadata']['appearance'] + 1) % 4
  np.testing.assert_array_equal(_np(model.apply({'params': fp}, gb, {})['fine']['rgb']), _np(a))


# ---------------------------------------------------------------------------------------------
# metadata_encoded=True (models.py:198-199, 210-211, 251; warping.py:378-381)
# ---------------------------------------------------------------------------------------------
def test_metadata_encoded_matches_ids_oracle_and_reference():
  r = _ref('nerf_encoded')
  spec = O.ModelSpec(num_coarse_samples=8, num_fine_samples=6, num_nerf_point_freqs=4, use_stratified_sampling=False, use_warp=True,
                     num_warp_freqs=5, num_warp_features=8, use_camera_metadata=True, use_appearance_metadata=True,
                     use_alpha_condition=True)
  seed = int(r['seed'])
  params = O.init_params(spec, seed=seed, trained_like=True)
  batch = O.synthetic_batch(3, seed=seed + 1)
  model, fp = H.gpu_model(spec, params, 3)
  gb = H.gpu_batch(batch)
  by_ids = model.apply({'params': fp}, gb, {'alpha': 3.25}, return_points=True, return_weights=True)
  enc = dict(gb)
  enc['metadata'] = {k: torch.tensor(r['codes/' + k]).float().to(DEV) for k in ('warp', 'appearance', 'camera')}
  by_codes = model.apply({'params': fp}, enc, {'alpha': 3.25}, metadata_encoded=True, return_points=True, return_weights=True,
                         return_warp_jacobian=True)
  for lv in ('coarse', 'fine'):
    for k in ('rgb', 'depth', 'acc', 'weights', 'warped_points'):
      np.testing.assert_allclose(_np(by_codes[lv][k]), _np(by_ids[lv][k]), atol=1e-6, err_msg=f'{lv}/{k}')     # same kernels, same codes
      np.testing.assert_allclose(_np(by_codes[lv][k]), r[f'{lv}/{k}'], atol=1e-4, err_msg=f'ref {lv}/{k}')     # the reference's own run
    np.testing.assert_allclose(_np(by_codes[lv]['warp_jacobian']), r[f'{lv}/warp_jacobian'], atol=2e-4)
  # interpolated codes (the video notebook's use of this switch): between two frames' renders, not equal to either
  mid = dict(enc)
  mid['metadata'] = {k: 0.5 * (v + v.roll(1, 0)) for k, v in enc['metadata'].items()}
  o = model.apply({'params': fp}, mid, {'alpha': 3.25}, metadata_encoded=True)
  ref_mid = O.nerf_model_apply(params, spec, {**batch, 'metadata': {k: v.double().cpu() for k, v in mid['metadata'].items()}}, 3.25,
                               metadata_encoded=True)
  np.testing.assert_allclose(_np(o['fine']['rgb']), _np(ref_mid['fine']['rgb']), atol=1e-4)
  from nerfies_amd.lib import NrfError
  with pytest.raises(NrfError):
    model.apply({'params': fp}, enc, {'alpha': 3.25}, metadata_encoded=True, train=True)
  with pytest.raises(NrfError):   # wrong code width
    bad = dict(enc); bad['metadata'] = dict(enc['metadata'], warp=enc['metadata']['warp'][:, :5])
    model.apply({'params': fp}, bad, {'alpha': 3.25}, metadata_encoded=True)


class _NearestKink:
  """oracle.relu_hook recorder: per sample, the smallest |pre-activation| / layer rms over every hidden unit."""

  def __init__(self):
    self.min = None

  def __call__(self, name, layer, pre):
    with torch.no_grad():
      d = pre.detach()
      r = (d.abs() / d.pow(2).mean().sqrt().clamp_min(1e-30)).reshape(-1, d.shape[-1]).min(-1).values
      self.min = r if self.min is None else torch.minimum(self.min, r)
    return torch.relu(pre)


# ---------------------------------------------------------------------------------------------
# return_warp_jacobian / use_warp_jacobian (models.py:264-265, 345-346, 367-368; warping.py:385-387)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('kw,alpha,B', [(dict(), 3.5, 7), (dict(num_warp_freqs=6, num_warp_features=3), 6.0, 40),
                                        (dict(warp_field_type='translation'), 2.0, 7),
                                        (dict(num_coarse_samples=128, num_fine_samples=128), 8.0, 96)])
def test_warp_jacobian_output(kw, alpha, B):
  spec = O.ModelSpec(**dict(dict(num_coarse_samples=32, num_fine_samples=32, use_warp=True), **kw))
  p = O.init_params(spec, seed=4, trained_like=True)
  b = O.synthetic_batch(B, seed=5)
  model, fp = H.gpu_model(spec, p, B)
  out = model.apply({'params': fp}, H.gpu_batch(b), {'alpha': alpha}, return_warp_jacobian=True, return_points=True)
  ref = O.nerf_model_apply(p, spec, b, alpha, return_warp_jacobian=True, return_points=True)
  for lv in ('coarse', 'fine'):
    J = out[lv]['warp_jacobian']
    assert tuple(J.shape) == (B, ref[lv]['z_vals'].shape[1], 3, 3)
    # the fine samples of the two sides differ by ~1e-6 in depth: compare J where the points agree, through the oracle
    # evaluated AT the GPU's points
    pts = out[lv]['points'].double().cpu()
    ids = b['metadata']['warp'][:, None, :].expand(B, pts.shape[1], 1)
    kink = _NearestKink()
    with O.relu_hook(kink):
      jo = O.se3_field(p['warp_field'], pts, ids, alpha, spec.num_warp_freqs, return_jacobian=True, name='jac/warp')['jacobian']
    # Float32 rounding of J scales with the sample's own |J| (the tangents carry the 2^(F_w-1) posenc factor): tolerance
    # 5e-5 + 2e-4 * max|J_sample|.  Beyond that, J is piecewise constant in the ReLU branch pattern: a sample with a trunk
    # unit within float32 rounding of its kink may sit on the other branch than float64 and differs by O(1) there.
    # Asserted: every sample deviating by more than the rounding tolerance has such a tie (|pre| < 2e-3 of its layer's
    # rms), and they are few.
    err = (J.double().cpu() - jo).abs().reshape(-1, 9).max(-1).values
    scale = jo.abs().reshape(-1, 9).max(-1).values
    bad = err > 5e-5 + 2e-4 * scale
    print(f'[jacobian {lv}] alpha {alpha}: err/scale median {(err / scale).median().item():.1e}, 99 % {(err / scale).quantile(0.99).item():.1e}; '
          f'{int(bad.sum())}/{bad.numel()} samples on another ReLU branch, their nearest kink <= {kink.min[bad].max().item() if bad.any() else 0:.1e}')
    assert bad.float().mean().item() < 0.06, (lv, bad.float().mean().item())
    assert (kink.min[bad] < 2e-3).all(), (lv, kink.min[bad].max().item())
    assert (err / scale).median().item() < 2e-5
    assert (J - torch.eye(3, device=DEV)).abs().max().item() > 1e-3      # a real deformation, not the identity
  # use_warp_jacobian on the model (construct_nerf(..., use_warp_jacobian=True)): coarse level only (models.py:345)
  model.use_warp_jacobian = True
  o2 = model.apply({'params': fp}, H.gpu_batch(b), {'alpha': alpha})
  assert 'warp_jacobian' in o2['coarse'] and 'warp_jacobian' not in o2['fine']
  np.testing.assert_allclose(_np(o2['coarse']['warp_jacobian']), _np(out['coarse']['warp_jacobian']), atol=1e-6)
  model.use_warp_jacobian = False
  o3 = model.apply({'params': fp}, H.gpu_batch(b), {'alpha': alpha})
  assert 'warp_jacobian' not in o3['coarse']
  np.testing.assert_allclose(_np(o3['fine']['rgb']), _np(out['fine']['rgb']), atol=1e-6)   # the extra pass changes nothing else


# ---------------------------------------------------------------------------------------------
# noise_std (model_utils.noise_regularize, model_utils.py:266-282)
# ---------------------------------------------------------------------------------------------
def test_noise_regularize_parity_and_philox():
  spec = O.ModelSpec(num_coarse_samples=32, num_fine_samples=32, use_stratified_sampling=True, noise_std=0.5)
  r = H.run_pinned(spec, 21, 0.0, seed=8)
  H.assert_pinned(r, 'noise_std=0.5 explicit normals')
  H.assert_forward(r, spec)
  model, fp, gb = r['model'], r['fp'], r['gb']
  # on-device Philox normals: reproducible per key, different across keys, and of the right scale (the rendered colour
  # moves by as much as with the explicit normals)
  base = {'coarse': r['rngs']['coarse'], 'fine': r['rngs']['fine']}
  a1 = model.apply({'params': fp}, gb, {}, rngs=base)['fine']['rgb'].clone()
  a2 = model.apply({'params': fp}, gb, {}, rngs=base)['fine']['rgb'].clone()
  np.testing.assert_array_equal(_np(a1), _np(a2))
  spec0 = O.ModelSpec(num_coarse_samples=32, num_fine_samples=32, use_stratified_sampling=True)
  m0, fp0 = H.gpu_model(spec0, O.init_params(spec, seed=8, trained_like=True), 21)
  clean = m0.apply({'params': fp0}, gb, {}, rngs=base)['fine']['rgb']
  d_philox = (a1 - clean).abs().mean().item()
  explicit = model.apply({'params': fp}, gb, {}, rngs=r['rngs'])['fine']['rgb']
  d_explicit = (explicit - clean).abs().mean().item()
  assert d_philox > 0 and 0.3 < d_philox / d_explicit < 3.0, (d_philox, d_explicit)
  # deterministic sampling switches the noise off (model_utils.py:278: only if use_stratified_sampling)
  spec_d = O.ModelSpec(num_coarse_samples=32, num_fine_samples=32, use_stratified_sampling=False, noise_std=0.5)
  md, fpd = H.gpu_model(spec_d, O.init_params(spec, seed=8, trained_like=True), 21)
  spec_c = O.ModelSpec(num_coarse_samples=32, num_fine_samples=32, use_stratified_sampling=False)
  mc, fpc = H.gpu_model(spec_c, O.init_params(spec, seed=8, trained_like=True), 21)
  np.testing.assert_array_equal(_np(md.apply({'params': fpd}, gb, {})['fine']['rgb']), _np(mc.apply({'params': fpc}, gb, {})['fine']['rgb']))


# ---------------------------------------------------------------------------------------------
# train_step's other loss terms (training.py:71-114, 199-222)
# ---------------------------------------------------------------------------------------------
WARP_KW = dict(num_coarse_samples=32, num_fine_samples=32, num_nerf_point_freqs=6, use_warp=True, use_stratified_sampling=True)


@pytest.mark.parametrize('ltype,method', [('svals', 'weight'), ('jtj', 'median'), ('div', 'weight'), ('det', 'weight'),
                                          ('log_det', 'median'), ('log_svals', 'median')])
def test_elastic_loss_types(ltype, method):
  spec = O.ModelSpec(**WARP_KW)
  r = H.run_pinned(spec, 12, 5.0, seed=13, elastic={'weight': 0.05, 'reduce_method': method, 'loss_type': ltype})
  H.assert_pinned(r, f'elastic {ltype}/{method}', loss_tol=2e-5)
  o, st = r['ostats']['coarse'], r['stats']
  assert o['loss/elastic'].item() > 0
  for i, k in ((6, 'loss/elastic'), (7, 'residual/elastic'), (12, 'metric/jacobian_det'), (13, 'metric/jacobian_div'), (14, 'metric/jacobian_curl')):
    assert abs(st[i].item() - o[k].item()) < 1e-6 + 3e-4 * abs(o[k].item()), (k, st[i].item(), o[k].item())


def test_unknown_elastic_type_is_rejected():
  from nerfies_amd.lib import NrfError
  spec = O.ModelSpec(**WARP_KW)
  model, fp = H.gpu_model(spec, O.init_params(spec, seed=1, trained_like=True), 4)
  gb = H.gpu_batch(O.synthetic_batch(4, seed=2))
  with pytest.raises(NrfError, match='nr'):
    model.loss_and_grad(fp, gb, warp_extra={'alpha': 1.0}, rngs={'coarse': 1, 'fine': 2}, elastic={'weight': 0.1, 'loss_type': 'nr'})


@pytest.mark.parametrize('kw,wr', [(dict(), dict(weight=0.5)), (dict(use_camera_metadata=True, num_warp_freqs=6), dict(weight=2.0, alpha=-2.0, scale=0.01)),
                                   (dict(warp_field_type='translation'), dict(weight=1.0))])
def test_warp_reg_loss(kw, wr):
  spec = O.ModelSpec(**dict(WARP_KW, **kw))
  r = H.run_pinned(spec, 14, 6.0, seed=17, warp_reg=wr)
  H.assert_pinned(r, f'warp_reg {kw}', loss_tol=3e-5)
  st = r['stats']
  for lv, i in (('coarse', 0), ('fine', 1)):
    o = r['ostats'][lv]
    assert o['loss/warp_reg'].item() > 0
    assert abs(st[8 + i].item() - o['loss/warp_reg'].item()) < 1e-7 + 3e-4 * o['loss/warp_reg'].item(), lv
    assert abs(st[10 + i].item() - o['residual/warp_reg'].item()) < 1e-7 + 3e-4 * o['residual/warp_reg'].item(), lv


def test_train_step_reports_every_reference_stat():
  """training.train_step with everything on: the stats dict carries the reference's keys (training.py:172-225, 259)."""
  from nerfies_amd import training
  spec = O.ModelSpec(**WARP_KW)
  model, fp = H.gpu_model(spec, O.init_params(spec, seed=3, trained_like=True), 16)
  gb = dict(H.gpu_batch(O.synthetic_batch(16, seed=4)))
  gb['background_points'] = (torch.rand(200, 3, device=DEV) - 0.5) * 0.8
  state = training.TrainState(optimizer=training.Optimizer(fp), warp_alpha=4.0)
  sp = training.ScalarParams(learning_rate=1e-3, elastic_loss_weight=0.01, warp_reg_loss_weight=0.1, background_loss_weight=1.0)
  state, stats, key = training.train_step(model, 5, state, gb, sp, use_elastic_loss=True, elastic_reduce_method='weight',
                                          elastic_loss_type='svals', use_background_loss=True, use_warp_reg_loss=True)
  assert set(stats['coarse']) == {'loss/rgb', 'loss/total', 'metric/psnr', 'loss/elastic', 'residual/elastic', 'loss/warp_reg',
                                  'residual/warp_reg', 'metric/jacobian_det', 'metric/jacobian_div', 'metric/jacobian_curl'}
  assert set(stats['fine']) == {'loss/rgb', 'loss/total', 'metric/psnr', 'loss/warp_reg', 'residual/warp_reg'}
  assert 'background_loss' in stats and all(torch.isfinite(v).all() for lv in ('coarse', 'fine') for v in stats[lv].values())
  total = stats['coarse']['loss/total'] + stats['fine']['loss/total'] + sp.background_loss_weight * stats['background_loss']
  assert abs(total.item() - state.optimizer.stats[4].item()) < 1e-5
  # stats are fresh tensors, not views of the donated gradient buffer (the next step must not change them)
  keep = stats['fine']['loss/rgb'].item()
  held = stats['fine']['loss/rgb']
  training.train_step(model, key, state, gb, sp, use_elastic_loss=True, elastic_reduce_method='weight', use_background_loss=True)
  assert held.item() == keep


# ---------------------------------------------------------------------------------------------
# 'time' warp metadata encoder: modules.TimeEncoder (modules.py:297-322) through SE3Field.encode_metadata
# (warping.py:256-259, 311-313) with metadata['time'] (models.py:252-254); csrc/time_encoder.hip
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('kw,B,alpha,time_alpha', [
    (dict(num_nerf_point_freqs=4, num_coarse_samples=16, num_fine_samples=16), 9, 3.0, 1.0),
    (dict(num_nerf_point_freqs=8, num_warp_freqs=6, num_warp_features=5, use_camera_metadata=True, num_coarse_samples=32, num_fine_samples=32), 33, 4.5, 0.4),
    (dict(num_nerf_point_freqs=6, warp_field_type='translation', num_coarse_samples=24, num_fine_samples=24), 12, 2.0, 0.0)])
def test_time_encoder_forward_and_gradients(kw, B, alpha, time_alpha):
  """Forward outputs and every gradient leaf -- the six TimeEncoder layers included -- against the float64 oracle (ReLU
  branches of the NeRF / warp trunks pinned to the HIP path's, tests/helpers.run_pinned)."""
  spec = O.ModelSpec(use_warp=True, use_stratified_sampling=True, warp_metadata_encoder_type='time', **kw)
  r = H.run_pinned(spec, B, alpha, seed=17, time_alpha=time_alpha)
  H.assert_pinned(r, f'time encoder B={B}')
  H.assert_forward(r, spec)
  enc = [k for k in r['errs'] if k.startswith('warp_field/metadata_encoder/mlp/')]
  assert len(enc) == 14 and all(r['errs'][k][1] > 0 for k in enc if k.endswith('kernel'))   # 6 hidden + logit, kernel + bias; all carry gradient
  assert not any('embed' in k for k in r['errs'] if k.startswith('warp_field'))              # no GLO table in this configuration


def test_time_encoder_codes_can_be_supplied_encoded():
  """metadata_encoded=True with the time encoder: the caller's codes replace the encoder's output (warping.py:378-381)."""
  spec = O.ModelSpec(use_warp=True, warp_metadata_encoder_type='time', num_coarse_samples=16, num_fine_samples=16)
  p = O.init_params(spec, seed=2, trained_like=True)
  b = O.synthetic_batch(10, seed=3)
  model, fp = H.gpu_model(spec, p, 10)
  gb = H.gpu_batch(b)
  by_time = model.apply({'params': fp}, gb, {'alpha': 3.0, 'time_alpha': 1.0})
  codes = O.time_encode(p['warp_field']['metadata_encoder'], b['metadata']['time'].double(), spec.num_time_encoder_freqs, 1.0)
  enc = dict(gb)
  enc['metadata'] = {'time': codes.float().to(DEV)}   # models.py:252-254: the warp metadata of a 'time' model is metadata['time']
  by_codes = model.apply({'params': fp}, enc, {'alpha': 3.0, 'time_alpha': 1.0}, metadata_encoded=True)
  np.testing.assert_allclose(_np(by_codes['fine']['rgb']), _np(by_time['fine']['rgb']), atol=2e-5)


# ---------------------------------------------------------------------------------------------
# one handle, several (num_rays, flags): the handle caches ONE workspace plan (include/nerfies_amd.h "Conventions")
# ---------------------------------------------------------------------------------------------
def test_one_handle_alternating_batch_sizes_flags_and_streams():
  """Calls on one handle are serialised by the caller, but they may alternate batch sizes, flags and streams freely: the plan is
  rebuilt and the descriptor tables are re-uploaded whenever (num_rays, flags) or the workspace change.  Interleaved inference at
  8 rays, training at 20 rays, bf16 inference at 8 rays -- the middle calls on a side stream -- must give exactly what fresh
  handles give for each call on its own."""
  spec = O.ModelSpec(num_coarse_samples=16, num_fine_samples=16, num_nerf_point_freqs=6, use_warp=True, num_warp_freqs=4)
  p = O.init_params(spec, seed=8, trained_like=True)
  b8, b20 = H.gpu_batch(O.synthetic_batch(8, seed=1)), H.gpu_batch(O.synthetic_batch(20, seed=2))
  we = {'alpha': 2.5}

  def fresh(fn):
    m, f = H.gpu_model(spec, p, 8)
    return fn(m, f)
  inf8 = lambda m, f: m.apply({'params': f}, b8, we)['fine']['rgb'].clone()
  bf8 = lambda m, f: m.apply({'params': f}, b8, we, bf16=True)['fine']['rgb'].clone()
  tr20 = lambda m, f: tuple(t.clone() for t in m.loss_and_grad(f, b20, warp_extra=we, rngs={'coarse': 3, 'fine': 4}))
  want8, want8b, (wantg, wants) = fresh(inf8), fresh(bf8), fresh(tr20)
  model, fp = H.gpu_model(spec, p, 8)
  side = torch.cuda.Stream()
  for rep in range(3):
    got8 = inf8(model, fp)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
      gotg, gots = tr20(model, fp)
      got8b = bf8(model, fp)
    side.synchronize()
    np.testing.assert_array_equal(_np(got8), _np(want8))
    np.testing.assert_array_equal(_np(got8b), _np(want8b))
    np.testing.assert_allclose(_np(gots), _np(wants), rtol=1e-6, atol=1e-7)
    assert (gotg - wantg).abs().max().item() <= 1e-5 * wantg.abs().max().item()   # embedding / per-ray sums use atomics


# ---------------------------------------------------------------------------------------------
# NerfMLP without any condition (use_viewdirs = False, no camera / appearance code): no bottleneck layer, the rgb branch reads
# the trunk output (modules.py:149-164); the library keeps its layer list and runs an internal identity in its place
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('kw,B,alpha', [
    (dict(num_nerf_point_freqs=6, num_coarse_samples=16, num_fine_samples=16), 40, 0.0),
    (dict(num_nerf_point_freqs=8, num_coarse_samples=32, num_fine_samples=32, use_warp=True, num_warp_freqs=4), 21, 3.0)])
def test_model_without_any_condition_forward_and_gradients(kw, B, alpha):
  spec = O.ModelSpec(use_stratified_sampling=True, use_viewdirs=False, **kw)
  r = H.run_pinned(spec, B, alpha, seed=23)
  H.assert_pinned(r, f'no condition B={B}')
  H.assert_forward(r, spec)
  names = [n for n, _, _ in r['model'].layout.entries]
  assert not any('bottleneck' in n for n in names)                       # the caller's tree is the reference's: no such leaf
  assert any(n.endswith('MLP_1/hidden_0/kernel') for n in names)
  # the bf16 mode runs the same internal identity (exact on bf16 values): loss and gradient direction follow the float32 path
  g32, s32 = r['model'].loss_and_grad(r['fp'], r['gb'], warp_extra={'alpha': alpha}, rngs=r['rngs'])
  g32, s32 = g32.clone(), s32.clone()
  g16, s16 = r['model'].loss_and_grad(r['fp'], r['gb'], warp_extra={'alpha': alpha}, rngs=r['rngs'], bf16='mlp')
  assert abs(s16[4].item() - s32[4].item()) < 1e-3 + 2e-2 * abs(s32[4].item())
  cos = torch.nn.functional.cosine_similarity(g16.double().flatten(), g32.double().flatten(), dim=0).item()
  assert cos > 0.98, cos


# ---------------------------------------------------------------------------------------------
# trunks shallower than the kernels' 8 layers (ModelConfig.nerf_trunk_depth; modules.MLP, modules.py:41-62): internal identity layers
# behind the caller's last one (relu(h . I) = h), zero posenc rows in layer 4 when the caller's trunk never reaches its skip
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('kw,B,alpha', [
    (dict(nerf_trunk_depth=6, num_nerf_point_freqs=6, num_coarse_samples=16, num_fine_samples=16, use_camera_metadata=True), 40, 0.0),
    (dict(nerf_trunk_depth=4, nerf_skips=(), num_nerf_point_freqs=8, num_coarse_samples=32, num_fine_samples=32, use_warp=True, num_warp_freqs=4), 21, 3.0),
    (dict(nerf_trunk_depth=2, num_nerf_point_freqs=4, num_coarse_samples=16, num_fine_samples=8, use_viewdirs=False), 17, 0.0)])
def test_shallow_trunk_forward_and_gradients(kw, B, alpha):
  spec = O.ModelSpec(use_stratified_sampling=True, **kw)
  r = H.run_pinned(spec, B, alpha, seed=29)
  H.assert_pinned(r, f'trunk depth {spec.nerf_trunk_depth} B={B}')
  H.assert_forward(r, spec)
  names = [n for n, _, _ in r['model'].layout.entries]
  assert sum('nerf_mlps_coarse/MLP_0/' in n and n.endswith('kernel') for n in names) == spec.nerf_trunk_depth
  g32, s32 = r['model'].loss_and_grad(r['fp'], r['gb'], warp_extra={'alpha': alpha}, rngs=r['rngs'])
  g32, s32 = g32.clone(), s32.clone()
  g16, s16 = r['model'].loss_and_grad(r['fp'], r['gb'], warp_extra={'alpha': alpha}, rngs=r['rngs'], bf16='mlp')
  assert abs(s16[4].item() - s32[4].item()) < 1e-3 + 2e-2 * abs(s32[4].item())
  cos = torch.nn.functional.cosine_similarity(g16.double().flatten(), g32.double().flatten(), dim=0).item()
  assert cos > 0.98, cos


# ---------------------------------------------------------------------------------------------
# round 6: nerf_skips at a layer other than 4 and ModelConfig.warp_kwargs trunk shapes (configs.py:63, 105; modules.py:47-48;
# warping.py:225-226, 90-91): forward + every gradient leaf against the pinned float64 oracle, the caller's tree shapes, and what the
# bfloat16 mode does with them
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('kw,B,alpha,bf16_ok', [
    (dict(nerf_skips=(5,), num_nerf_point_freqs=6, num_coarse_samples=16, num_fine_samples=16, use_camera_metadata=True), 40, 0.0, False),
    (dict(nerf_skips=(7,), num_nerf_point_freqs=4, num_coarse_samples=16, num_fine_samples=8, use_warp=True, num_warp_freqs=4), 19, 2.0, False),
    (dict(nerf_skips=(1,), num_nerf_point_freqs=8, num_coarse_samples=24, num_fine_samples=24, use_warp=True, num_warp_freqs=5), 21, 3.0, False),
    (dict(nerf_skips=(2,), nerf_trunk_depth=6, num_nerf_point_freqs=6, num_coarse_samples=16, num_fine_samples=16, use_warp=True, num_warp_freqs=4), 23, 2.5, True),
    (dict(nerf_skips=(3,), nerf_trunk_depth=5, num_nerf_point_freqs=4, num_coarse_samples=16, num_fine_samples=8, use_viewdirs=False), 17, 0.0, True)])
def test_moved_skip_forward_and_gradients(kw, B, alpha, bf16_ok):
  from nerfies_amd import lib as L
  spec = O.ModelSpec(use_stratified_sampling=True, **kw)
  r = H.run_pinned(spec, B, alpha, seed=31)
  H.assert_pinned(r, f'nerf_skips {spec.nerf_skips} depth {spec.nerf_trunk_depth} B={B}')
  H.assert_forward(r, spec)
  shapes = {n: tuple(sh) for n, _, sh in r['model'].layout.entries}
  s, P = spec.nerf_skips[0], 3 + 6 * spec.num_nerf_point_freqs
  for lv in ('coarse', 'fine'):
    for i in range(spec.nerf_trunk_depth):
      want = (P if i == 0 else 256) + (P if i == s else 0)
      assert shapes[f'nerf_mlps_{lv}/MLP_0/hidden_{i}/kernel'] == (want, 256), (lv, i, shapes[f'nerf_mlps_{lv}/MLP_0/hidden_{i}/kernel'])
    assert f'nerf_mlps_{lv}/MLP_0/hidden_{spec.nerf_trunk_depth}/kernel' not in shapes
  g32, s32 = r['model'].loss_and_grad(r['fp'], r['gb'], warp_extra={'alpha': alpha}, rngs=r['rngs'])
  g32, s32 = g32.clone(), s32.clone()
  if bf16_ok:    # laid out around the chains' own layer 4: the bf16 stream runs it
    g16, s16 = r['model'].loss_and_grad(r['fp'], r['gb'], warp_extra={'alpha': alpha}, rngs=r['rngs'], bf16='mlp')
    assert abs(s16[4].item() - s32[4].item()) < 1e-3 + 2e-2 * abs(s32[4].item())
    cos = torch.nn.functional.cosine_similarity(g16.double().flatten(), g32.double().flatten(), dim=0).item()
    assert cos > 0.98, cos
  else:          # a moved skip GEMM exists only in the float32 chains: refused, not silently run at layer 4
    with pytest.raises(L.NrfError, match='bfloat16|BF16'):
      r['model'].loss_and_grad(r['fp'], r['gb'], warp_extra={'alpha': alpha}, rngs=r['rngs'], bf16='mlp')


@pytest.mark.parametrize('kw,B,alpha', [
    (dict(warp_trunk_depth=5, warp_trunk_width=96, num_warp_freqs=5), 21, 3.25),
    (dict(warp_trunk_depth=3, warp_trunk_width=64, num_warp_freqs=4), 19, 1.5),
    (dict(warp_trunk_depth=1, warp_trunk_width=128, num_warp_freqs=6), 17, 4.0),
    (dict(warp_trunk_depth=6, warp_trunk_width=40, num_warp_freqs=4, use_camera_metadata=True), 17, 2.0),
    (dict(warp_trunk_depth=4, warp_trunk_width=80, num_warp_freqs=5, warp_field_type='translation'), 20, 2.25)])
def test_warp_kwargs_trunk_shapes_forward_and_gradients(kw, B, alpha):
  """SE3Field(trunk_depth, trunk_width) / TranslationField(depth, hidden_channels) from ModelConfig.warp_kwargs, with the elastic and
  background regularisers on (the Jacobian tangents and the background warp run the same padded trunk)."""
  spec = O.ModelSpec(use_stratified_sampling=True, use_warp=True, num_nerf_point_freqs=6, num_coarse_samples=16, num_fine_samples=16, **kw)
  g = torch.Generator().manual_seed(5)
  nbg = 37
  bg = {'points': torch.rand(nbg, 3, generator=g).double() - 0.5, 'warp_ids': torch.randint(0, spec.num_warp_embeddings, (nbg,), generator=g),
        'noise': 0.001 * torch.randn(nbg, 3, generator=g).double(), 'weight': 1.0}
  se3 = spec.warp_field_type == 'se3'
  r = H.run_pinned(spec, B, alpha, seed=37, elastic={'weight': 0.01, 'reduce_method': 'weight'} if se3 else None, background=bg)
  H.assert_pinned(r, f'warp trunk {spec.warp_trunk_depth} x {spec.warp_trunk_width} ({spec.warp_field_type}) B={B}')
  H.assert_forward(r, spec)
  shapes = {n: tuple(sh) for n, _, sh in r['model'].layout.entries}
  trunk = 'warp_field/trunk' if se3 else 'warp_field/mlp'
  Win, W = 3 + 6 * spec.num_warp_freqs + spec.num_warp_features, spec.warp_trunk_width
  for i in range(spec.warp_trunk_depth):
    assert shapes[f'{trunk}/hidden_{i}/kernel'] == ((Win if i == 0 else W) + (Win if i == 4 else 0), W)
  assert f'{trunk}/hidden_{spec.warp_trunk_depth}/kernel' not in shapes
  assert shapes['warp_field/branches_v/logit/kernel' if se3 else 'warp_field/mlp/logit/kernel'] == (W, 3)
  # the bf16 warp trunk runs the same padded image
  g32, s32 = r['model'].loss_and_grad(r['fp'], r['gb'], warp_extra={'alpha': alpha}, rngs=r['rngs'])
  g32, s32 = g32.clone(), s32.clone()
  g16, s16 = r['model'].loss_and_grad(r['fp'], r['gb'], warp_extra={'alpha': alpha}, rngs=r['rngs'], bf16=True)
  assert abs(s16[4].item() - s32[4].item()) < 1e-3 + 3e-2 * abs(s32[4].item())
  cos = torch.nn.functional.cosine_similarity(g16.double().flatten(), g32.double().flatten(), dim=0).item()
  assert cos > 0.9, cos     # measured 0.947 (one-layer trunk: the heads read bf16 roundings of relu(W0 posenc) directly) .. 0.99


def test_unsupported_warp_kwargs_are_refused_by_name():
  from nerfies_amd import lib as L
  for bad in (dict(use_pivot=True), dict(use_translation=True), dict(rotation_depth=2), dict(trunk_depth=7), dict(trunk_width=256),
              dict(skips=(2,)), dict(min_freq_log2=1)):
    spec = O.ModelSpec(use_warp=True)
    cfg = H.config_from_spec(spec)
    cfg.warp_kwargs = bad
    from nerfies_amd import models
    with pytest.raises(L.NrfError, match='warp_kwargs'):
      models.construct_nerf(0, cfg, 8, [0, 1], [0, 1], [0, 1, 2, 3], spec.near, spec.far)
  # defaults spelled out are accepted
  cfg = H.config_from_spec(O.ModelSpec(use_warp=True))
  cfg.warp_kwargs = dict(trunk_depth=6, trunk_width=128, skips=(4,), use_pivot=False, rotation_width=64)
  models.construct_nerf(0, cfg, 8, [0, 1], [0, 1], [0, 1, 2, 3], spec.near, spec.far)
