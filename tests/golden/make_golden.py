#!/usr/bin/env python
"""Generates tests/golden/*.npz from the fp64 CPU oracle (oracle/nerfies_oracle.py).

The reference (google/nerfies) cannot be imported here (JAX/Flax absent) and ships no golden
vectors for this path, so these fixtures freeze the ORACLE's outputs ("parity unpinned", DESIGN.md
section 2): they catch regressions of the oracle itself (tests/test_golden.py, CPU) and are the
fixed targets of the GPU parity tests.  Parameters are NOT stored (1.2 M floats per case): they are
regenerated from the seed with O.init_params, which is deterministic numpy.

  python tests/golden/make_golden.py        # rewrites every fixture
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

from oracle import nerfies_oracle as O  # noqa: E402

# name -> (ModelSpec kwargs, num_rays, sampling mode, warp_alpha)
CASES = {
    # BASELINE configs[1] shape (gpu_quarterhd as measured): 64+128 samples, F_p=8, warp off
    'quarterhd_det': (dict(num_coarse_samples=64, num_fine_samples=128, num_nerf_point_freqs=8,
                           use_stratified_sampling=False), 6, 'det', 0.0),
    'quarterhd_strat': (dict(num_coarse_samples=64, num_fine_samples=128, num_nerf_point_freqs=8,
                             use_stratified_sampling=True), 6, 'uniforms', 0.0),
    # BASELINE configs[0] plumbing shape (test_local.gin): 64+64, F_p=10, warp off here
    'plumbing_nowarp': (dict(num_coarse_samples=64, num_fine_samples=64, num_nerf_point_freqs=10,
                             use_stratified_sampling=False), 5, 'det', 0.0),
    # camera-conditioned rgb branch (gpu_vrig_paper.gin: use_camera_metadata, R = 27 + 2)
    'camera_cond': (dict(num_coarse_samples=32, num_fine_samples=32, num_nerf_point_freqs=8,
                         use_stratified_sampling=False, use_camera_metadata=True, num_camera_embeddings=2), 5, 'det', 0.0),
    # SE3 warp on (test_local.gin: F_w=8, G=3 -> here G=8 preset value), half-annealed window
    'warp_se3': (dict(num_coarse_samples=32, num_fine_samples=32, num_nerf_point_freqs=8,
                      use_stratified_sampling=False, use_warp=True, num_warp_freqs=8, num_warp_features=8,
                      num_warp_embeddings=4), 5, 'det', 3.5),
    # gpu_vrig_paper.gin: F_w=6, camera code, stratified
    'warp_vrig': (dict(num_coarse_samples=32, num_fine_samples=32, num_nerf_point_freqs=8,
                       use_stratified_sampling=True, use_warp=True, num_warp_freqs=6, num_warp_features=8,
                       num_warp_embeddings=4, use_camera_metadata=True, num_camera_embeddings=2), 5, 'uniforms', 6.0),
    # every loss term of gpu_vrig_paper.gin: SE3 warp + elastic ('weight') + background regulariser, camera code.
    # Low NeRF posenc frequencies keep the fp32/fp64 ReLU branches identical (see compute()).
    'vrig_full': (dict(num_coarse_samples=32, num_fine_samples=32, num_nerf_point_freqs=2, use_stratified_sampling=True,
                       use_warp=True, num_warp_freqs=6, num_warp_features=8, num_warp_embeddings=4,
                       use_camera_metadata=True, num_camera_embeddings=2), 5, 'uniforms', 6.0),
}
# loss kwargs of the cases that train with regularisers
LOSS_KW = {'vrig_full': dict(use_elastic_loss=True, elastic_loss_weight=0.01, elastic_reduce_method='weight',
                             use_background_loss=True, background_loss_weight=1.0)}


def background_inputs(name, spec):
  seed = sum(ord(c) for c in name) + 7
  rng = np.random.default_rng(seed)
  n = 150
  return {'points': torch.tensor(rng.uniform(-0.4, 0.4, size=(n, 3))), 'warp_ids': torch.tensor(rng.integers(0, spec.num_warp_embeddings, size=(n, 1))),
          'noise': torch.tensor(1e-3 * rng.normal(size=(n, 3)))}


def case_inputs(name):
  kw, B, mode, alpha = CASES[name]
  spec = O.ModelSpec(**kw)
  seed = sum(ord(c) for c in name)
  params = O.init_params(spec, seed=seed, trained_like=True, dtype=torch.float64)
  batch = O.synthetic_batch(B, seed=seed + 1, dtype=torch.float64)
  rng = np.random.default_rng(seed + 2)
  t_rand = u = None
  if mode == 'uniforms':
    t_rand = torch.tensor(rng.uniform(0, 1, size=(B, spec.num_coarse_samples)).astype(np.float32)).double()
    u = torch.tensor(rng.uniform(0, 1, size=(B, spec.num_fine_samples)).astype(np.float32)).double()
  return spec, params, batch, t_rand, u, alpha


def leaf_digest(t):
  """(sum, abs-sum, first, last, max-abs) of a gradient leaf: enough to pin it without storing it."""
  f = t.reshape(-1)
  return np.array([f.sum().item(), f.abs().sum().item(), f[0].item(), f[-1].item(), f.abs().max().item()])


def compute(name):
  spec, params, batch, t_rand, u, alpha = case_inputs(name)
  lkw = dict(LOSS_KW.get(name, {}))
  if lkw.get('use_background_loss'):
    lkw['background'] = background_inputs(name, spec)
  total, stats, grads, ret = O.loss_and_grad(params, spec, batch, warp_alpha=alpha, t_rand=t_rand, u=u, **lkw)
  out = {'loss': np.array(total.item())}
  if 'background' in lkw:
    out['background_loss'] = np.array(stats['background_loss'].item())
    for k, v in lkw['background'].items():
      out['in/background/' + k] = v.numpy()
  if lkw.get('use_elastic_loss'):
    out['coarse/loss_elastic'] = np.array(stats['coarse']['loss/elastic'].item())
    out['coarse/residual_elastic'] = np.array(stats['coarse']['residual/elastic'].item())
  for lv in ret:
    for k in ('rgb', 'depth', 'med_depth', 'acc', 'weights', 'z_vals'):
      out[f'{lv}/{k}'] = ret[lv][k].detach().numpy()
    out[f'{lv}/mse'] = np.array(stats[lv]['loss/rgb'].item())
    out[f'{lv}/psnr'] = np.array(stats[lv]['metric/psnr'].item())
  for path, g in O.tree_leaves_with_path(grads):
    out['grad/' + path] = leaf_digest(g)
  # The same gradients from the fp32 oracle.  A pre-activation within fp32 rounding of zero takes
  # the other ReLU branch in fp32 than in fp64; with few rays one such flip moves a whole layer's
  # gradient (and every layer below it) by percents.  An fp32 implementation is therefore held to
  # the fp32 oracle where the two oracles disagree (tests/test_golden.py).
  f32 = lambda t: t.float() if torch.is_tensor(t) and t.is_floating_point() else t
  p32 = O.tree_map(f32, params)
  b32 = {k: (O.tree_map(f32, v) if isinstance(v, dict) else f32(v)) for k, v in batch.items()}
  lkw32 = dict(lkw)
  if 'background' in lkw32:
    lkw32['background'] = {k: f32(v) for k, v in lkw32['background'].items()}
  _, _, grads32, _ = O.loss_and_grad(p32, spec, b32, warp_alpha=alpha, t_rand=f32(t_rand) if t_rand is not None else None,
                                     u=f32(u) if u is not None else None, **lkw32)
  for path, g in O.tree_leaves_with_path(grads32):
    out['grad32/' + path] = leaf_digest(g.double())
  for k in ('origins', 'directions', 'rgb'):
    out['in/' + k] = batch[k].numpy()
  for k, v in batch['metadata'].items():
    if k != 'time':   # only read by the TimeEncoder variant, which has no fixture here
      out['in/metadata/' + k] = v.numpy()
  if t_rand is not None:
    out['in/t_rand'] = t_rand.numpy()
    out['in/u'] = u.numpy()
  return out


def main():
  for name in CASES:
    out = compute(name)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print(f'{name}: loss {float(out["loss"]):.6f}  -> {path} ({os.path.getsize(path)} bytes)')


if __name__ == '__main__':
  main()
