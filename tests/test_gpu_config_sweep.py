"""Randomised configuration sweep: forward outputs and the loss of the HIP path against the fp64 oracle over the
option space nrf_create accepts (sample counts, encoder frequencies, widths, every boolean switch, both warp field
types, ragged batch sizes).  Seeded, so a failure is reproducible from its case index."""
import numpy as np
import pytest
import torch

import helpers as H
from oracle import nerfies_oracle as O

pytestmark = pytest.mark.gpu


def _case(i):
  rng = np.random.default_rng(1000 + i)
  pick = lambda *xs: xs[int(rng.integers(len(xs)))]
  use_warp = bool(rng.integers(2))
  use_viewdirs = bool(rng.integers(4) > 0)
  kw = dict(
      num_coarse_samples=int(pick(3, 8, 17, 32, 64, 96)), num_fine_samples=int(pick(1, 5, 16, 33, 64)),
      num_nerf_point_freqs=int(pick(1, 4, 8, 10)), num_nerf_viewdir_freqs=int(pick(0, 2, 4)),
      use_viewdirs=use_viewdirs, use_camera_metadata=bool(rng.integers(2)) or not use_viewdirs,
      use_appearance_metadata=bool(rng.integers(2)), use_stratified_sampling=bool(rng.integers(2)),
      use_white_background=bool(rng.integers(2)), use_linear_disparity=bool(rng.integers(2)),
      use_sample_at_infinity=bool(rng.integers(2)), sigma_activation=pick('softplus', 'relu'),
      nerf_trunk_width=int(pick(256, 256, 128, 96)), nerf_rgb_branch_width=int(pick(128, 128, 64)),
      use_warp=use_warp)
  if use_warp:
    kw.update(warp_field_type=pick('se3', 'translation'), num_warp_freqs=int(pick(0, 3, 6, 8)), num_warp_features=int(pick(1, 4, 8)))
  B = int(pick(1, 5, 31, 64, 70))
  return kw, B, float(rng.uniform(0, 8))


@pytest.mark.parametrize('i', range(int(__import__('os').environ.get('SWEEP_FROM', 0)), int(__import__('os').environ.get('SWEEP_TO', 24))))
def test_random_configuration(i):
  kw, B, alpha = _case(i)
  spec = O.ModelSpec(**kw)
  oparams = O.init_params(spec, seed=i, trained_like=True, dtype=torch.float64)
  batch = O.synthetic_batch(B, seed=100 + i, dtype=torch.float64)
  model, fp = H.gpu_model(spec, oparams, B)
  gb = H.gpu_batch(batch)
  rngs, t_rand, u = None, None, None
  if spec.use_stratified_sampling:
    g = torch.Generator().manual_seed(i)
    t_rand = torch.rand(B, spec.num_coarse_samples, generator=g)
    u = torch.rand(B, spec.num_fine_samples, generator=g)
    rngs = {'coarse': t_rand.to(H.DEV), 'fine': u.to(H.DEV)}
    t_rand, u = t_rand.double(), u.double()
  out = model.apply({'params': fp}, gb, {'alpha': alpha}, rngs=rngs, return_weights=True)
  ref = O.nerf_model_apply(oparams, spec, batch, alpha, t_rand=t_rand, u=u)
  # posenc at 2^9 amplifies the fp32 rounding of the sample positions; the bound stays inside the north star's 1e-3
  tol = 1e-4 if spec.num_nerf_point_freqs <= 8 else 4e-4
  if spec.use_warp:
    tol *= 3
  for lv in ('coarse', 'fine'):
    for k in ('rgb', 'acc', 'weights'):
      err = (out[lv][k].cpu().double() - ref[lv][k]).abs().max().item()
      # a single sample's weight moves more than the composited colour when an inverse-CDF sample shifts by an fp32 ulp
      assert err < (3 * tol if k == 'weights' else tol), (i, kw, lv, k, err)
    derr = (out[lv]['depth'].cpu().double() - ref[lv]['depth']).abs().max().item()
    assert derr < tol * (1.0 if not spec.use_linear_disparity else 2.0), (i, kw, lv, 'depth', derr)
  grad, stats = model.loss_and_grad(fp, gb, warp_extra={'alpha': alpha}, rngs=rngs)
  zf = model.apply({'params': fp}, gb, {'alpha': alpha}, rngs=rngs, return_z_vals=True)['fine']['z_vals'].cpu().double()
  loss, _, ograds, _ = O.loss_and_grad(oparams, spec, batch, warp_alpha=alpha, t_rand=t_rand, u=u, fixed_fine_z=zf)
  assert abs(stats[4].item() - loss.item()) < 20 * tol * max(1.0, abs(loss.item())), (i, kw, stats[4].item(), loss.item())
  assert torch.isfinite(grad).all()
  # gradient: every leaf's norm against the oracle's (leaf-by-leaf element parity is tests/test_gpu_parity.py's job; a relu
  # sigma can legitimately be dead for a whole batch, in which case both sides are exactly zero)
  from nerfies_amd import params as P
  got = P.tree_from_flat(grad.cpu(), model.layout)
  gmax = max(t.abs().max().item() for _, t in O.tree_leaves_with_path(ograds))
  for path, og in O.tree_leaves_with_path(ograds):
    node = got
    for k in path.split('/'):
      node = node[k]
    a, b = node.double().norm().item(), og.norm().item()
    assert abs(a - b) <= 0.05 * max(a, b) + 1e-3 * gmax + 1e-12, (i, kw, path, a, b)
