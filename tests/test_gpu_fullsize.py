"""Size-independent properties at BASELINE.json's full sizes: config A (1024 rays x (64+128) samples, warp off) and the
gpu_vrig_paper shape (768 rays x (128+128), SE3 warp).  Every property below holds for the reference by construction (rays
are independent, the loss is a mean, the VJP is linear).  The oracle-backed parity tests at small sizes are in
tests/test_gpu_parity.py; since round 2 the float64 oracle itself is ALSO run at these full shapes (and config D's),
tests/test_gpu_pinned.py -- the properties here stay as the cheap, oracle-free cross-check."""
import numpy as np
import pytest
import torch

import helpers as H
from oracle import nerfies_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _setup(kind, seed=0):
  if kind == 'A':
    B = 1024
    spec = O.ModelSpec(num_coarse_samples=64, num_fine_samples=128, num_nerf_point_freqs=8, use_stratified_sampling=False)
  else:
    B = 768
    spec = O.ModelSpec(num_coarse_samples=128, num_fine_samples=128, num_nerf_point_freqs=8, use_stratified_sampling=False,
                       use_warp=True, num_warp_freqs=6, use_camera_metadata=True)
  oparams = O.init_params(spec, seed=seed, trained_like=True, dtype=torch.float32)
  batch = O.synthetic_batch(B, seed=seed + 1, dtype=torch.float32)
  model, fp = H.gpu_model(spec, oparams, B)
  return spec, model, fp, H.gpu_batch(batch), B


def _take(batch, idx):
  out = {k: v[idx] for k, v in batch.items() if torch.is_tensor(v)}
  out['metadata'] = {k: v[idx] for k, v in batch['metadata'].items()}
  return out


@pytest.mark.parametrize('kind', ['A', 'vrig'])
def test_forward_invariants_and_ray_independence(kind):
  spec, model, fp, gb, B = _setup(kind)
  extra = {'alpha': 4.0}
  out = model.apply({'params': fp}, gb, extra, return_weights=True, return_z_vals=True)
  for lv, S in (('coarse', spec.num_coarse_samples), ('fine', spec.num_coarse_samples + spec.num_fine_samples)):
    o = out[lv]
    assert o['weights'].shape == (B, S) and o['rgb'].shape == (B, 3)
    assert torch.isfinite(o['rgb']).all() and (o['rgb'] >= 0).all() and (o['rgb'] <= 1 + 1e-6).all()
    assert (o['weights'] >= 0).all()
    # sample_at_infinity: the last sample (dist 1e10) absorbs the remaining transmittance and is left out of acc
    # (model_utils.py:124-127)
    np.testing.assert_allclose(o['weights'][:, :-1].sum(-1).cpu().numpy(), o['acc'].cpu().numpy(), atol=5e-6)
    assert (o['weights'].sum(-1) <= 1 + 1e-4).all() and (o['acc'] <= 1 + 1e-5).all()
    z = o['z_vals']
    assert (z[:, 1:] >= z[:, :-1]).all() and (z >= spec.near - 1e-6).all() and (z <= spec.far + 1e-6).all()
    # expected depth lies between the first and last sample that carry weight (sample_at_infinity: last dist = 1e10)
    assert (o['depth'] >= 0).all() and (o['med_depth'] >= z[:, 0] - 1e-6).all() and (o['med_depth'] <= z[:, -1] + 1e-6).all()
  # rays are independent units: a permutation of the batch permutes the outputs exactly, a sub-batch reproduces its rows
  g = torch.Generator().manual_seed(5)
  perm = torch.randperm(B, generator=g).to(DEV)
  outp = model.apply({'params': fp}, _take(gb, perm), extra)
  for lv in ('coarse', 'fine'):
    for k in ('rgb', 'depth', 'acc', 'med_depth'):
      assert torch.equal(outp[lv][k], out[lv][k][perm]), (lv, k)
  sub = torch.arange(100, 100 + 193, device=DEV)            # ragged: not a multiple of the 64-row tile
  outs = model.apply({'params': fp}, _take(gb, sub), extra)
  assert torch.equal(outs['fine']['rgb'], out['fine']['rgb'][sub])
  again = model.apply({'params': fp}, gb, extra)
  assert torch.equal(again['fine']['rgb'], out['fine']['rgb']) and torch.equal(again['coarse']['depth'], out['coarse']['depth'])


@pytest.mark.parametrize('kind', ['A', 'vrig'])
def test_vjp_is_linear_and_shards_add_up(kind):
  spec, model, fp, gb, B = _setup(kind, seed=2)
  extra = {'alpha': 4.0}
  g = torch.Generator().manual_seed(3)
  d1c, d1f, d2c, d2f = (torch.randn(B, 3, generator=g).to(DEV) / B for _ in range(4))
  model.apply({'params': fp}, gb, extra, train=True)
  g1 = model.backward({'params': fp}, gb, d1c, d1f).clone()
  g2 = model.backward({'params': fp}, gb, d2c, d2f).clone()
  g12 = model.backward({'params': fp}, gb, 0.5 * d1c - 2.0 * d2c, 0.5 * d1f - 2.0 * d2f).clone()
  want = 0.5 * g1 - 2.0 * g2
  scale = want.abs().max().item()
  assert scale > 0 and (g12 - want).abs().max().item() < 5e-5 * scale
  # zero upstream gradient on half of the rays == the gradient of the other half alone (scaled batches are independent)
  half = torch.zeros(B, 1, device=DEV); half[: B // 2] = 1.0
  ga = model.backward({'params': fp}, gb, d1c * half, d1f * half).clone()
  first = torch.arange(B // 2, device=DEV)
  sb = _take(gb, first)
  model.apply({'params': fp}, sb, extra, train=True)
  gb_half = model.backward({'params': fp}, sb, d1c[: B // 2], d1f[: B // 2]).clone()
  assert (ga - gb_half).abs().max().item() < 5e-5 * max(ga.abs().max().item(), 1e-12)


@pytest.mark.parametrize('kind', ['A', 'vrig'])
@pytest.mark.parametrize('level', ['coarse', 'fine'])
def test_loss_gradient_matches_a_directional_finite_difference(kind, level):
  """(L(p + e v) - L(p - e v)) / 2e == <grad, v> at full size, v a random direction over ONE level's MLP leaves.
  The fine sample depths are a stop_gradient function of the coarse weights (models.py:353-357), so a finite
  difference is only comparable where it cannot move them: fine-MLP directions against the total loss, coarse-MLP
  directions against the coarse loss term (the fine term depends on the coarse MLP through the sampling only)."""
  spec, model, fp, gb, B = _setup(kind, seed=4)
  extra = {'alpha': 4.0}
  grad, stats = model.loss_and_grad(fp, gb, warp_extra=extra)
  grad = grad.clone()
  g = torch.Generator().manual_seed(9)
  v = torch.zeros_like(fp.flat)
  for name, off, shape in model.layout.entries:
    if not name.startswith(f'nerf_mlps_{level}/'):
      continue
    n = int(np.prod(shape))
    leaf = fp.flat[off:off + n]
    v[off:off + n] = torch.randn(n, generator=g).to(DEV) * max(leaf.abs().mean().item(), 1e-3)
  eps = 3e-4
  base = fp.flat.clone()
  target = gb['rgb'].double()

  def loss_at(t):   # the losses re-formed in float64 from the rendered colours: the library's fp32 sums are too noisy for a difference
    fp.flat.copy_(base + t * v)
    out = model.apply({'params': fp}, gb, extra)
    mse = {lv: float(((out[lv]['rgb'].double() - target) ** 2).mean()) for lv in ('coarse', 'fine')}
    return mse['coarse'] + mse['fine'] if level == 'fine' else mse['coarse']
  fd = (loss_at(eps) - loss_at(-eps)) / (2 * eps)
  fp.flat.copy_(base)
  an = float((grad.double() * v.double()).sum())
  print(f'[fd] {kind} {level}: fd={fd:.6e} analytic={an:.6e}')
  assert abs(an) > 1e-4 and abs(fd - an) < 0.03 * max(abs(an), abs(fd)) + 5e-6, (fd, an)
