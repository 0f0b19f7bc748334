"""Known-answer checks that pin the CPU oracle (SURVEY.md 8c i-ix).

The reference ships no tests or golden vectors for this path and cannot be
imported here (no JAX/Flax), so the oracle is pinned against independent
formulations instead: scipy.linalg.expm, np.searchsorted, sequential fp64
loops, closed forms and finite differences.
"""
import math

import numpy as np
import pytest
import scipy.linalg
import torch

from oracle import nerfies_oracle as O

torch.manual_seed(0)
F64 = torch.float64


def test_exp_se3_matches_expm():
  rng = np.random.default_rng(1)
  for _ in range(20):
    w = rng.normal(size=3)
    w /= np.linalg.norm(w)
    v = rng.normal(size=3)
    theta = rng.uniform(0.01, 2.5)
    T = O.exp_se3(torch.tensor(np.concatenate([w, v])), torch.tensor(theta, dtype=F64)).numpy()
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    M = np.zeros((4, 4))
    M[:3, :3] = W * theta
    M[:3, 3] = v * theta
    np.testing.assert_allclose(T, scipy.linalg.expm(M), atol=1e-12)


def test_se3_closed_form_matches_matrix_form():
  """x' = x + A(wxx) + B wx(wxx) + v + B(wxv) + C wx(wxv)."""
  rng = np.random.default_rng(2)
  w = torch.tensor(rng.normal(size=(50, 3)) * 0.3)
  v = torch.tensor(rng.normal(size=(50, 3)) * 0.3)
  x = torch.tensor(rng.normal(size=(50, 3)))
  theta = torch.linalg.norm(w, dim=-1)
  T = O.exp_se3(torch.cat([w / theta[:, None], v / theta[:, None]], -1), theta)
  ref = O.from_homogenous((T @ O.to_homogenous(x)[..., None])[..., 0])
  A = torch.sin(theta) / theta
  B = (1 - torch.cos(theta)) / theta ** 2
  C = (theta - torch.sin(theta)) / theta ** 3
  cr = torch.linalg.cross
  wx, wv = cr(w, x), cr(w, v)
  out = (x + A[:, None] * wx + B[:, None] * cr(w, wx) + v + B[:, None] * wv +
         C[:, None] * cr(w, wv))
  np.testing.assert_allclose(out.numpy(), ref.numpy(), atol=1e-13)


def test_sinusoidal_encoder_order_and_values():
  x = torch.tensor([[0.3, -0.7, 1.1]], dtype=F64)
  F = 5
  enc = O.sinusoidal_encode(x, F).numpy()[0]
  assert enc.shape == (3 + 6 * F,)
  np.testing.assert_allclose(enc[:3], x[0].numpy())
  for k in range(F):
    for c in range(3):
      a = (2.0 ** k) * x[0, c].item()
      np.testing.assert_allclose(enc[3 + (2 * k) * 3 + c], math.sin(a), atol=1e-14)
      np.testing.assert_allclose(enc[3 + (2 * k + 1) * 3 + c], math.cos(a), atol=1e-14)


def test_sinusoidal_encoder_fp32_slack_vs_true_cos():
  """fp32 sin(x + fp32(pi/2)) vs cos: documents the ulp-level slack."""
  x = torch.rand(1000, 3) * 2 - 1
  enc = O.sinusoidal_encode(x, 8)
  ref = O.sinusoidal_encode(x.double(), 8)
  assert (enc.double() - ref).abs().max() < 2e-5  # |angle|<=128 -> ~128*2^-24


def test_cosine_easing_window():
  F = 8
  assert torch.allclose(O.cosine_easing_window(F, 0.0, F64), torch.zeros(F, dtype=F64))
  assert torch.allclose(O.cosine_easing_window(F, float(F), F64), torch.ones(F, dtype=F64))
  w = O.cosine_easing_window(F, 3.5, F64)
  np.testing.assert_allclose(w.numpy(), [1, 1, 1, 0.5, 0, 0, 0, 0], atol=1e-15)


def test_annealed_encoder_windows_bands():
  x = torch.randn(7, 3, dtype=F64)
  full = O.sinusoidal_encode(x, 6)
  np.testing.assert_allclose(O.annealed_sinusoidal_encode(x, 6, 6.0).numpy(), full.numpy(), atol=1e-15)
  half = O.annealed_sinusoidal_encode(x, 6, 2.0)
  np.testing.assert_allclose(half[:, :3 + 12].numpy(), full[:, :3 + 12].numpy(), atol=1e-15)
  assert half[:, 3 + 12:].abs().max() < 1e-15


def test_mlp_skip_concat_order():
  """Hand-built 2-layer MLP proving concat([h, inputs]) (modules.py:47-48)."""
  p = {'hidden_0': {'kernel': torch.eye(2, dtype=F64), 'bias': torch.zeros(2, dtype=F64)},
       'hidden_1': {'kernel': torch.tensor([[1.], [0.], [0.], [10.]], dtype=F64),
                    'bias': torch.zeros(1, dtype=F64)}}
  x = torch.tensor([[2., 3.]], dtype=F64)
  # h = [2,3]; concat([h, x]) = [2,3,2,3]; out = 2*1 + 3*10 = 32
  assert O.mlp(p, x, 2, (1,), False).item() == 32.0


def _render_loop(rgb, sigma, z, d):
  """Sequential fp64 per-sample compositing (independent formulation)."""
  B, S = sigma.shape
  out = np.zeros((B, 3)); depth = np.zeros(B); acc = np.zeros(B); med = np.zeros(B)
  W = np.zeros((B, S))
  for b in range(B):
    T = 1.0
    nd = np.linalg.norm(d[b])
    cum = 0.0
    found = False
    for i in range(S):
      dist = (z[b, i + 1] - z[b, i]) if i + 1 < S else 1e10
      a = 1.0 - math.exp(-sigma[b, i] * dist * nd)
      w = a * T
      W[b, i] = w
      out[b] += w * rgb[b, i]
      depth[b] += w * z[b, i]
      if i < S - 1:
        acc[b] += w
      cum += w
      if not found and cum >= 0.5:
        med[b] = z[b, i]
        found = True
      T *= (1.0 - a + 1e-10)
  return out, depth, med, acc, W


def test_volumetric_rendering_vs_loop():
  rng = np.random.default_rng(3)
  B, S = 6, 33
  rgb = rng.uniform(size=(B, S, 3)); sigma = rng.uniform(0, 30, size=(B, S))
  sigma[0] = 0.0          # never crosses 0.5 except through the sample at infinity
  z = np.sort(rng.uniform(0.1, 1.0, size=(B, S)), -1); d = rng.normal(size=(B, 3))
  o = O.volumetric_rendering(torch.tensor(rgb), torch.tensor(sigma), torch.tensor(z),
                             torch.tensor(d), False)
  ref = _render_loop(rgb, sigma, z, d)
  for got, want in zip((o['rgb'], o['depth'], o['med_depth'], o['acc'], o['weights']), ref):
    np.testing.assert_allclose(got.numpy(), want, atol=1e-12)
  # sigma == 0 everywhere: only the 1e10 last sample could fire, and alpha there is 0 too
  assert o['acc'][0] == 0 and o['med_depth'][0] == 0


def test_volumetric_rendering_constant_sigma_closed_form():
  S, s = 50, 3.0
  z = torch.linspace(0.2, 1.2, S, dtype=F64)[None]
  d = torch.tensor([[0., 0., 1.]], dtype=F64)
  o = O.volumetric_rendering(torch.ones(1, S, 3, dtype=F64), torch.full((1, S), s, dtype=F64), z, d, False)
  dz = (z[0, 1] - z[0, 0]).item()
  # sum_{i<S-1} alpha T^i with T = exp(-s dz) (+1e-10) -> 1 - exp(-s (S-1) dz)
  np.testing.assert_allclose(o['acc'].item(), 1 - math.exp(-s * (S - 1) * dz), rtol=1e-7)
  np.testing.assert_allclose(o['rgb'].numpy(), np.ones((1, 3)), rtol=1e-7)


def _pdf_searchsorted(bins, weights, u):
  """SURVEY A.5 formulation with np.searchsorted(side='right')."""
  w = weights + 1e-5
  pdf = w / w.sum(-1, keepdims=True)
  cdf = np.concatenate([np.zeros_like(pdf[..., :1]), np.cumsum(pdf, -1)], -1)
  n = bins.shape[-1]
  out = np.zeros_like(u)
  for b in range(bins.shape[0]):
    idx = np.searchsorted(cdf[b], u[b], side='right')
    lo = np.clip(idx - 1, 0, n - 2); hi = np.clip(idx, 1, n - 1)
    den = cdf[b, hi] - cdf[b, lo]
    den = np.where(den < 1e-5, 1.0, den)
    out[b] = bins[b, lo] + (u[b] - cdf[b, lo]) / den * (bins[b, hi] - bins[b, lo])
  return out


def test_piecewise_constant_pdf_vs_searchsorted():
  rng = np.random.default_rng(4)
  B, n, N = 5, 63, 128
  bins = np.sort(rng.uniform(0.1, 1.0, size=(B, n)), -1)
  w = rng.uniform(size=(B, n - 1)) ** 4
  w[1] = 0.0
  u = rng.uniform(size=(B, N)); u[0, 0] = 0.0; u[0, 1] = 1.0 - 1e-16
  got = O.piecewise_constant_pdf(torch.tensor(bins), torch.tensor(w), N, True, torch.tensor(u))
  np.testing.assert_allclose(got.numpy(), _pdf_searchsorted(bins, w, u), atol=1e-13)
  # deterministic: u = linspace(0,1,N), includes u == 1 exactly
  got = O.piecewise_constant_pdf(torch.tensor(bins), torch.tensor(w), N, False)
  ul = np.broadcast_to(np.linspace(0., 1., N), (B, N))
  np.testing.assert_allclose(got.numpy(), _pdf_searchsorted(bins, w, ul), atol=1e-13)
  assert (np.diff(got.numpy(), axis=-1) >= -1e-15).all()   # sorted => sort() is a merge


def test_pdf_uniform_weights_gives_linspace():
  n, N = 17, 33
  bins = torch.linspace(1.0, 2.0, n, dtype=F64)[None]
  got = O.piecewise_constant_pdf(bins, torch.ones(1, n - 1, dtype=F64), N, False)
  np.testing.assert_allclose(got[0].numpy(), np.linspace(1.0, 2.0, N), atol=1e-12)


def test_general_loss_geman_mcclure_closed_form():
  x2 = torch.rand(100, dtype=F64) * 0.01
  c = 0.03
  s = x2 / c ** 2
  np.testing.assert_allclose(
      O.general_loss_with_squared_residual(x2, -2.0, c).numpy(),
      (c * 2 * s / (s + 4)).numpy(), atol=1e-15)


def test_elastic_svals_vs_eig_jtj():
  J = torch.eye(3, dtype=F64) + 0.2 * torch.randn(10, 3, 3, dtype=F64)
  loss, res = O.compute_elastic_loss(J)
  ev = torch.linalg.eigvalsh(J.transpose(-1, -2) @ J)
  r2 = (0.5 * torch.log(ev)) ** 2
  np.testing.assert_allclose((res ** 2).numpy(), r2.sum(-1).numpy(), atol=1e-12)


def test_adam_first_step_is_lr_sign():
  p = torch.zeros(4, dtype=F64); g = torch.tensor([1., -2., 0.5, 3.], dtype=F64)
  p1, m, v = O.adam_update(p, torch.zeros_like(p), torch.zeros_like(p), g, 0, 1e-3)
  np.testing.assert_allclose(p1.numpy(), -1e-3 * np.sign(g.numpy()), rtol=1e-6)


def test_adam_matches_an_independent_implementation_over_steps():
  """flax.optim.Adam is absent here; torch.optim.Adam implements the same published rule (bias-corrected moments,
  eps added to sqrt(v_hat)) independently: ten steps with changing gradients and a changing learning rate."""
  g = torch.Generator().manual_seed(0)
  p0 = torch.randn(50, dtype=F64, generator=g)
  ref_p = p0.clone().requires_grad_(True)
  opt = torch.optim.Adam([ref_p], lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
  p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
  for step in range(10):
    grad = torch.randn(50, dtype=F64, generator=g) * (1.0 + step)
    lr = 1e-3 * 0.9 ** step
    for group in opt.param_groups:
      group['lr'] = lr
    ref_p.grad = grad.clone()
    opt.step()
    p, m, v = O.adam_update(p, m, v, grad, step, lr)
    np.testing.assert_allclose(p.numpy(), ref_p.detach().numpy(), rtol=0, atol=1e-14)


def _small_spec(**kw):
  base = dict(num_coarse_samples=8, num_fine_samples=8, nerf_trunk_width=16,
              nerf_rgb_branch_width=8, num_nerf_point_freqs=3, num_nerf_viewdir_freqs=2,
              num_warp_freqs=3, num_warp_features=4)
  base.update(kw)
  return O.ModelSpec(**base)


@pytest.mark.parametrize('use_warp', [False, True])
def test_gradients_vs_finite_differences(use_warp):
  spec = _small_spec(use_warp=use_warp, use_camera_metadata=True)
  params = O.init_params(spec, seed=1, trained_like=True)
  batch = O.synthetic_batch(5, seed=2)
  kw = dict(warp_alpha=2.5)
  loss, _, grads, ret = O.loss_and_grad(params, spec, batch, **kw)
  # FD must see the same stop_gradient on the fine samples as autodiff does.
  kw['fixed_fine_z'] = ret['fine']['z_vals'].detach()
  gl = dict(O.tree_leaves_with_path(grads))
  pl = dict(O.tree_leaves_with_path(params))
  rng = np.random.default_rng(0)
  names = ['nerf_mlps_coarse/MLP_0/hidden_4/kernel', 'nerf_mlps_fine/MLP_1/hidden_0/kernel',
           'nerf_mlps_fine/MLP_2/logit/kernel', 'camera_encoder/embed/embedding']
  if use_warp:
    names += ['warp_field/trunk/hidden_4/kernel', 'warp_field/branches_w/logit/kernel',
              'warp_field/branches_v/logit/bias', 'warp_field/metadata_encoder/embed/embedding']
  for name in names:
    t = pl[name]
    for _ in range(3):
      idx = tuple(rng.integers(0, s) for s in t.shape)
      old = t[idx].item()
      h = 1e-6
      t[idx] = old + h
      lp = O.loss_fn(params, spec, batch, **kw)[0].item()
      t[idx] = old - h
      lm = O.loss_fn(params, spec, batch, **kw)[0].item()
      t[idx] = old
      fd = (lp - lm) / (2 * h)
      assert abs(fd - gl[name][idx].item()) < 1e-6 + 1e-4 * abs(fd), (name, idx)


def test_no_gradient_from_fine_into_coarse_mlp():
  """stop_gradient on z_samples (model_utils.py:187): fine MSE has zero grad
  w.r.t. the coarse MLP."""
  spec = _small_spec()
  params = O.init_params(spec, seed=3, trained_like=True)
  batch = O.synthetic_batch(4, seed=4)
  leaves = [t.requires_grad_(True) for _, t in O.tree_leaves_with_path(params)]
  ret = O.nerf_model_apply(params, spec, batch)
  fine_loss = ((ret['fine']['rgb'] - batch['rgb']) ** 2).mean()
  g = torch.autograd.grad(fine_loss, leaves, allow_unused=True)
  for (name, _), gi in zip(O.tree_leaves_with_path(params), g):
    if name.startswith('nerf_mlps_coarse'):
      assert gi is None or gi.abs().max() == 0


def test_warp_jacobian_vs_finite_differences():
  spec = _small_spec(use_warp=True)
  params = O.init_params(spec, seed=5, trained_like=True)
  pts = torch.randn(4, 3, 3, dtype=F64) * 0.3
  ids = torch.randint(0, 4, (4, 3, 1))
  out = O.se3_field(params['warp_field'], pts, ids, 2.2, spec.num_warp_freqs, True)
  J = out['jacobian']
  h = 1e-6
  for c in range(3):
    e = torch.zeros(3, dtype=F64); e[c] = h
    fp = O.se3_field(params['warp_field'], pts + e, ids, 2.2, spec.num_warp_freqs)['warped_points']
    fm = O.se3_field(params['warp_field'], pts - e, ids, 2.2, spec.num_warp_freqs)['warped_points']
    np.testing.assert_allclose(J[..., c].detach().numpy(), ((fp - fm) / (2 * h)).detach().numpy(), atol=1e-7)


def test_fine_pass_evaluates_all_sorted_samples():
  spec = _small_spec()
  params = O.init_params(spec, seed=6)
  batch = O.synthetic_batch(3, seed=7)
  ret = O.nerf_model_apply(params, spec, batch)
  zf = ret['fine']['z_vals']
  assert zf.shape == (3, 16)
  assert (zf[:, 1:] >= zf[:, :-1]).all()
  zc = ret['coarse']['z_vals']
  for b in range(3):
    for zi in zc[b]:
      assert (zf[b] == zi).any()
