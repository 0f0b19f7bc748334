"""Gradient parity of the HIP path against the fp64 oracle at the configurations that are actually trained:
NeRF posenc F_p = 8 / 10 with the SE3 warp on, full BASELINE batch shapes (configs A, C, D), and a 20-step
Adam trajectory.

Why "pinned": with the warp on, the float32 rounding of a warped point (~1e-7 relative) is amplified by 2^(F_p-1)
in the posenc angle, so a handful of trunk pre-activations that sit within rounding of zero take the other ReLU
branch than in float64 -- and than in ANY other float32 evaluation order.  One flipped unit moves the gradient of
its whole weight column (and of every layer below) by an O(1/rows) fraction, i.e. by percents at 5-30 rays.  Round 1
therefore held these cases to a 20 % magnitude digest, which is exactly where a real bug could hide.  Here the
ambiguity is removed instead: the HIP path's own ReLU sign bits are read back from the training workspace
(nrf_debug_ws_offset "bits_trunk" / "bits_rgbh" / "w_bits"), the fp64 oracle is evaluated with its hidden
activations pinned to that branch pattern (oracle.relu_hook: activation = pre * mask), and every leaf must then agree
to 2e-3 of its max-abs entry (measured ~1e-5).  The test also asserts that the pattern the oracle would have chosen
itself differs from the HIP one only in a tiny fraction of units and only where |pre| is at rounding level -- so the
pinning changes nothing but the tie-breaks.

Tolerances (SURVEY 8d / north star): rendered rgb/depth/acc <= 1e-4 abs (north star: 1e-3), loss <= 1e-5,
per-leaf gradients <= 2e-3 of the leaf's max-abs.
"""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, 'golden'))

from oracle import nerfies_oracle as O  # noqa: E402
import helpers as H  # noqa: E402
from helpers import run_pinned, assert_pinned, assert_forward, host_threads as _threads, leaf as _leaf  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'

# ---------------------------------------------------------------------------------------------
# (a) warp on at the presets' posenc widths, small batches: the cases round 1 could only hold to 20 %
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('kw,B,alpha', [
    (dict(num_nerf_point_freqs=8, num_coarse_samples=32, num_fine_samples=32), 9, 3.5),
    (dict(num_nerf_point_freqs=10, num_coarse_samples=32, num_fine_samples=32), 9, 8.0),
    (dict(num_nerf_point_freqs=8, num_warp_freqs=6, use_camera_metadata=True, num_coarse_samples=48, num_fine_samples=48), 24, 6.0),
    (dict(num_nerf_point_freqs=10, num_warp_features=3, num_coarse_samples=64, num_fine_samples=64), 16, 1.25),
    (dict(num_nerf_point_freqs=8, warp_field_type='translation', num_coarse_samples=32, num_fine_samples=32), 9, 6.0),
])
def test_warp_gradients_at_preset_posenc_width(kw, B, alpha):
  spec = O.ModelSpec(use_warp=True, use_stratified_sampling=True, **kw)
  r = run_pinned(spec, B, alpha)
  assert_pinned(r, f'warp F_p={spec.num_nerf_point_freqs} B={B}')
  assert_forward(r, spec)
  got_embed = r['errs']['warp_field/metadata_encoder/embed/embedding']
  assert got_embed[1] > 0   # the GLO table carries gradient


@pytest.mark.parametrize('name', ['warp_se3', 'warp_vrig', 'vrig_full'])
def test_golden_warp_cases_pinned(name):
  """The golden warp fixtures (tests/golden/make_golden.py inputs) to 2e-3 per leaf -- replaces the 20 % digest."""
  import make_golden as G
  spec, params, batch, t_rand, u, alpha = G.case_inputs(name)
  lkw = G.LOSS_KW.get(name, {})
  bg = el = None
  if lkw.get('use_background_loss'):
    bg = dict(G.background_inputs(name, spec), weight=lkw['background_loss_weight'])
  if lkw.get('use_elastic_loss'):
    el = {'weight': lkw['elastic_loss_weight'], 'reduce_method': lkw['elastic_reduce_method']}
  r = run_pinned(spec, batch['origins'].shape[0], alpha, params=params, batch=batch, t_rand=t_rand, u=u, elastic=el, background=bg)
  assert_pinned(r, name, loss_tol=3e-5)


# ---------------------------------------------------------------------------------------------
# (b) full BASELINE shapes against the oracle (fp64 on the host cores: tens of seconds each)
# ---------------------------------------------------------------------------------------------
def test_config_a_full_batch_vs_oracle():
  """configs[1]: 1024 rays x (64+128), F_p=8, warp off, stratified -- the headline bench shape, every workgroup of the
  persistent chain kernels runs several tiles."""
  spec = O.ModelSpec(num_coarse_samples=64, num_fine_samples=128, num_nerf_point_freqs=8, use_stratified_sampling=True)
  with _threads(64):
    r = run_pinned(spec, 1024, 0.0, seed=11)
  assert_pinned(r, 'config A 1024x(64+128)')
  assert_forward(r, spec)
  with _threads(64):
    H.assert_unpinned(r, 'config A 1024x(64+128)')


def test_config_c_full_shard_vs_oracle():
  """configs[2] per-GPU shard: 768 rays x (128+128), SE3 warp F_w=6 + camera code + elastic ('weight') + background."""
  spec = O.ModelSpec(num_coarse_samples=128, num_fine_samples=128, num_nerf_point_freqs=8, use_stratified_sampling=True,
                     use_warp=True, num_warp_freqs=6, num_warp_features=8, use_camera_metadata=True)
  rng = np.random.default_rng(5)
  nbg = 2048
  bg = {'points': torch.tensor(rng.uniform(-0.4, 0.4, size=(nbg, 3))), 'warp_ids': torch.tensor(rng.integers(0, 4, size=(nbg, 1))),
        'noise': torch.tensor(1e-3 * rng.normal(size=(nbg, 3))), 'weight': 1.0}
  with _threads(64):
    r = run_pinned(spec, 768, 6.0, seed=12, elastic={'weight': 0.01, 'reduce_method': 'weight'}, background=bg)
  assert_pinned(r, 'config C 768x(128+128) warp+elastic+bg', loss_tol=2e-5)
  o = r['ostats']
  assert abs(r['stats'][5].item() - o['background_loss'].item()) < 1e-6 + 2e-4 * abs(o['background_loss'].item())
  assert abs(r['stats'][6].item() - o['coarse']['loss/elastic'].item()) < 1e-6 + 2e-4 * abs(o['coarse']['loss/elastic'].item())
  with _threads(64):
    H.assert_unpinned(r, 'config C 768x(128+128) warp+elastic+bg')


def test_config_d_full_shard_vs_oracle():
  """configs[3] per-GPU shard: 512 rays x (256+256), F_p=10, SE3 warp F_w=8 on (fp32 path; the bf16 mode has its own
  dPSNR gate in tests/test_gpu_bf16.py)."""
  spec = O.ModelSpec(num_coarse_samples=256, num_fine_samples=256, num_nerf_point_freqs=10, use_stratified_sampling=True,
                     use_warp=True, num_warp_freqs=8, num_warp_features=8)
  with _threads(64):
    r = run_pinned(spec, 512, 8.0, seed=13)
  assert_pinned(r, 'config D 512x(256+256) F_p=10 warp')
  assert_forward(r, spec)
  with _threads(64):
    H.assert_unpinned(r, 'config D 512x(256+256) F_p=10 warp')


# ---------------------------------------------------------------------------------------------
# (c) training trajectory: 20 Adam steps, GPU vs oracle, same init and the same uniforms every step
# ---------------------------------------------------------------------------------------------
def _oracle_trajectory(spec, p0, batch, uniforms, lr, dtype):
  cast = lambda t: t.to(dtype) if torch.is_tensor(t) and t.is_floating_point() else t
  b = {k: (O.tree_map(cast, v) if isinstance(v, dict) else cast(v)) for k, v in batch.items()}
  leaves = [(path, t.to(dtype).clone()) for path, t in O.tree_leaves_with_path(p0)]
  m = [torch.zeros_like(t) for _, t in leaves]
  v = [torch.zeros_like(t) for _, t in leaves]
  losses = []
  for k, (t_rand, u) in enumerate(uniforms):
    it = iter([t for _, t in leaves])
    cur = O.tree_map(lambda _: next(it), p0)
    loss, _, grads, _ = O.loss_and_grad(cur, spec, b, t_rand=t_rand.to(dtype), u=u.to(dtype))
    losses.append(loss.item())
    for j, (_, gt) in enumerate(O.tree_leaves_with_path(grads)):
      pnew, m[j], v[j] = O.adam_update(leaves[j][1], m[j], v[j], gt, k, lr)
      leaves[j] = (leaves[j][0], pnew)
  return np.array(losses), dict(leaves)


def test_training_trajectory_matches_oracle():
  """training.train_step x 20 (training.py:138-271; flax Adam defaults) on 64 rays x (64+128), warp off, against the
  float64 oracle from the same init with the same uniforms.

  What CAN be asserted.  Adam normalises every entry by its own gradient history (m / sqrt(v)), so an entry whose
  gradient is small against its leaf's max -- where float32 has few correct digits -- gets an update that is percents
  off, in ANY float32 implementation, and the loss landscape amplifies parameter differences step by step: the oracle's
  own float32 run ends 2e-4 (relative L2 per leaf, 1 % of the distance travelled, max entry 3e-3) from its float64 run at
  lr = 1e-4, and its loss curve 4e-4 away at lr = 1e-3.  Element-wise 1e-4 agreement of the parameters is therefore not
  a property a float32 path can have.  Asserted here, at lr = 1e-4: loss curve within 1e-5 of the float64 oracle's, every
  leaf within 1e-3 relative L2 of it, and no further from it than 3x the float32 oracle's own distance (+ 1e-5) --
  i.e. the HIP path tracks the reference as closely as float32 arithmetic allows.  Single-step gradients and the Adam
  kernel are pinned separately (2e-3 / 2e-6)."""
  from nerfies_amd import params as P, training
  B, K, lr = 64, 20, 1e-4
  spec = O.ModelSpec(num_coarse_samples=64, num_fine_samples=128, num_nerf_point_freqs=8, use_stratified_sampling=True)
  p64 = O.init_params(spec, seed=21, trained_like=True, dtype=torch.float64)
  b64 = O.synthetic_batch(B, seed=22, dtype=torch.float64)
  g = torch.Generator().manual_seed(23)
  uniforms = [(torch.rand(B, spec.num_coarse_samples, generator=g), torch.rand(B, spec.num_fine_samples, generator=g)) for _ in range(K)]
  model, fp = H.gpu_model(spec, p64, B)
  gb = H.gpu_batch(b64)
  state = training.TrainState(optimizer=training.Optimizer(fp))
  sp = training.ScalarParams(learning_rate=lr)
  gpu_loss = []
  for k, (t_rand, u) in enumerate(uniforms):
    state, stats, _ = training.train_step(model, k, state, gb, sp, rngs={'coarse': t_rand.to(DEV), 'fine': u.to(DEV)})
    gpu_loss.append((stats['coarse']['loss/rgb'] + stats['fine']['loss/rgb']).item())
  with _threads(32):
    l64, w64 = _oracle_trajectory(spec, p64, b64, uniforms, lr, torch.float64)
    l32, w32 = _oracle_trajectory(spec, p64, b64, uniforms, lr, torch.float32)
  np.testing.assert_allclose(gpu_loss, l64, atol=1e-5)
  assert l64[-1] < l64[0]
  got = P.tree_from_flat(fp.flat.cpu(), model.layout)
  worst = (0.0, 0.0, 0.0)
  for path, want in w64.items():
    have = _leaf(got, path).double()
    nrm = max(want.norm().item(), 1e-30)
    l2_gpu, l2_f32 = (have - want).norm().item() / nrm, (w32[path].double() - want).norm().item() / nrm
    mx = (have - want).abs().max().item() / max(want.abs().max().item(), 1e-30)
    worst = max(worst, (l2_gpu, l2_f32, mx))
    assert l2_gpu < 1e-3, (path, l2_gpu)
    assert l2_gpu < 3 * l2_f32 + 1e-5, (path, l2_gpu, l2_f32)
  print(f'[trajectory] {K} steps at lr {lr}: loss max dev gpu {np.abs(np.array(gpu_loss) - l64).max():.1e} (float32 oracle '
        f'{np.abs(l32 - l64).max():.1e}); worst leaf rel-L2 gpu {worst[0]:.1e} (float32 oracle {worst[1]:.1e}), max entry {worst[2]:.1e}')
