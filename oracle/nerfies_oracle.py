"""CPU oracle for the nerfies hot path.  TEST INFRASTRUCTURE ONLY.

PARITY PINNED against the reference's own sources: JAX/Flax are not installable in
the build container, so the reference cannot run natively, but its modules
(rigid_body, model_utils, modules, warping, models, training, utils) are imported
UNMODIFIED from /root/reference and executed on NumPy float64 through import-name
stand-ins (oracle/_shim: jax.numpy -> numpy, a minimal eager flax.linen,
jax.random -> supplied arrays, jax.jacfwd -> central differences).  Their outputs
are committed as tests/golden/ref_*.npz (tests/golden/make_reference_vectors.py)
and this restatement reproduces them to 1e-8..1e-12, including NerfModel.apply end
to end with the warp field, SE3Field / TranslationField with their Jacobians,
compute_elastic_loss and compute_background_loss (tests/test_reference_vectors.py).  Not covered by that
route: reverse-mode gradients (no autodiff in the shim) -- those are torch.autograd
on the pinned forward, checked by finite differences
(tests/test_oracle_known_answers.py) -- and XLA's float32 evaluation order.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  The product path (nerfies_amd/) never does.

Every function restates the reference formulas in the op order of the cited
lines (all paths relative to /root/reference/nerfies/).  Tensors are torch CPU
tensors; dtype follows the inputs (float64 = ground truth, float32 = the timed
"reference restated on CPU" baseline).  Gradients come from torch.autograd on
this restatement (the reference uses jax.value_and_grad, training.py:264-265).
"""
from __future__ import annotations

import math
from typing import Any, Dict, Optional, Sequence, Tuple

import numpy as np
import torch

Tensor = torch.Tensor


# ----------------------------------------------------------------------------
# rigid_body.py
# ----------------------------------------------------------------------------
def skew(w: Tensor) -> Tensor:
  """rigid_body.py:21-36.  w (...,3) -> (...,3,3) with skew(w) @ v == w x v."""
  z = torch.zeros_like(w[..., 0])
  return torch.stack([
      torch.stack([z, -w[..., 2], w[..., 1]], -1),
      torch.stack([w[..., 2], z, -w[..., 0]], -1),
      torch.stack([-w[..., 1], w[..., 0], z], -1),
  ], -2)


def exp_so3(w: Tensor, theta: Tensor) -> Tensor:
  """rigid_body.py:54-68 (Rodrigues)."""
  W = skew(w)
  eye = torch.eye(3, dtype=w.dtype).expand(W.shape)
  th = theta[..., None, None]
  return eye + torch.sin(th) * W + (1.0 - torch.cos(th)) * (W @ W)


def exp_se3(S: Tensor, theta: Tensor) -> Tensor:
  """rigid_body.py:71-89 (Modern Robotics 3.88).  S (...,6) -> (...,4,4)."""
  w, v = S[..., :3], S[..., 3:]
  W = skew(w)
  R = exp_so3(w, theta)
  eye = torch.eye(3, dtype=S.dtype).expand(W.shape)
  th = theta[..., None, None]
  p = (th * eye + (1.0 - torch.cos(th)) * W +
       (th - torch.sin(th)) * (W @ W)) @ v[..., None]
  top = torch.cat([R, p], -1)
  bottom = torch.zeros_like(top[..., :1, :])
  bottom[..., 0, 3] = 1.0
  return torch.cat([top, bottom], -2)  # rp_to_se3, rigid_body.py:39-51


def to_homogenous(v: Tensor) -> Tensor:
  return torch.cat([v, torch.ones_like(v[..., :1])], -1)  # rigid_body.py:92-93


def from_homogenous(v: Tensor) -> Tensor:
  return v[..., :3] / v[..., -1:]  # rigid_body.py:96-97


# ----------------------------------------------------------------------------
# modules.py
# ----------------------------------------------------------------------------
def sinusoidal_encode(x: Tensor, num_freqs: int, scale: float = 1.0) -> Tensor:
  """modules.py:172-228.  x (...,C) -> (..., C + 2*F*C).

  freqs = 2**linspace(0, F-1, F); features (F,2,C) = sin(stack(angles,
  angles + pi/2)); identity prepended.  cos is evaluated as sin(a + pi/2) with
  pi/2 rounded to the working dtype (modules.py:219-223).
  """
  if num_freqs == 0:
    return x
  freqs = 2.0 ** torch.linspace(0.0, num_freqs - 1.0, num_freqs, dtype=x.dtype)
  angles = scale * x[..., None, :] * freqs[:, None]           # (...,F,C)
  half_pi = torch.tensor(math.pi / 2, dtype=x.dtype)
  feats = torch.stack((angles, angles + half_pi), -2)         # (...,F,2,C)
  feats = torch.sin(feats.reshape(*x.shape[:-1], -1))
  return torch.cat([x, feats], -1)


def cosine_easing_window(num_bands: int, alpha, dtype) -> Tensor:
  """modules.py:274-294."""
  bands = torch.linspace(0.0, num_bands - 1.0, num_bands, dtype=dtype)
  a = torch.as_tensor(alpha, dtype=dtype)
  x = torch.clip(a - bands, 0.0, 1.0)
  return 0.5 * (1 + torch.cos(math.pi * x + math.pi))


def annealed_sinusoidal_encode(x: Tensor, num_freqs: int, alpha) -> Tensor:
  """modules.py:231-272: window each band's sin/cos pair, identity untouched."""
  if num_freqs == 0:
    return x
  C = x.shape[-1]
  feats = sinusoidal_encode(x, num_freqs)
  ident, feats = feats[..., :C], feats[..., C:]
  feats = feats.reshape(*x.shape[:-1], num_freqs, 2, C)
  window = cosine_easing_window(num_freqs, alpha, x.dtype).reshape(-1, 1, 1)
  feats = (window * feats).reshape(*x.shape[:-1], -1)
  return torch.cat([ident, feats], -1)


# Test hook on every Dense: None = the reference's arithmetic.  A callable (p, x) -> y lets a test restate a REDUCED-PRECISION
# mode of the HIP path operand for operand (tests/test_gpu_bf16_train.py: bfloat16 rounding of activations, weights and of
# the back-propagated pre-activation gradients, exactly where csrc/mlp_bf16.hip rounds them).
_DENSE_HOOK = None
_SCOPE = None   # 'nerf_mlp' while nerf_mlp() runs: lets a dense hook tell the NeRF MLPs from the warp field's


class dense_hook:
  """with dense_hook(fn): ... -- every dense(p, x) call returns fn(p, x)."""

  def __init__(self, fn):
    self.fn = fn

  def __enter__(self):
    global _DENSE_HOOK
    self.prev, _DENSE_HOOK = _DENSE_HOOK, self.fn
    return self

  def __exit__(self, *exc):
    global _DENSE_HOOK
    _DENSE_HOOK = self.prev
    return False


def dense(p: Dict[str, Tensor], x: Tensor) -> Tensor:
  """flax nn.Dense: y = x @ kernel[in,out] + bias (modules.py:42-58)."""
  if _DENSE_HOOK is not None:
    return _DENSE_HOOK(p, x)
  return x @ p['kernel'] + p['bias']


# Test hook on the hidden activations: None = nn.relu.  A callable (name, layer, pre) -> activation lets a test
# (a) record the pre-activations and (b) PIN the ReLU branch pattern to one measured elsewhere
# (activation = pre * mask, derivative = mask): a pre-activation within float32 rounding of zero takes the other
# branch in float32 than in float64, which moves whole gradient columns by percents at small batch sizes; with the
# pattern pinned to the HIP path's own sign bits the comparison is exact again (tests/test_gpu_pinned.py).
# `name` = '<level>/MLP_0' (NeRF trunk), '<level>/MLP_1' (rgb branch), '<level>/warp' (SE3 / translation trunk),
# level in {coarse, fine, background}.
_RELU_HOOK = None


class relu_hook:
  """with relu_hook(fn): ... -- installs `fn(name, layer, pre)` as the hidden activation of every mlp() call."""

  def __init__(self, fn):
    self.fn = fn

  def __enter__(self):
    global _RELU_HOOK
    self.prev, _RELU_HOOK = _RELU_HOOK, self.fn
    return self

  def __exit__(self, *exc):
    global _RELU_HOOK
    _RELU_HOOK = self.prev
    return False


def mlp(p: Dict[str, Any], x: Tensor, depth: int, skips: Sequence[int],
        has_logit: bool, name: Optional[str] = None) -> Tensor:
  """modules.py:26-62.  ReLU after every hidden layer; skip = concat([x, inputs])."""
  inputs = x
  for i in range(depth):
    if i in skips:
      x = torch.cat([x, inputs], -1)
    pre = dense(p[f'hidden_{i}'], x)
    x = torch.relu(pre) if (_RELU_HOOK is None or name is None) else _RELU_HOOK(name, i, pre)
  if has_logit:
    x = dense(p['logit'], x)
  return x


def nerf_mlp(p: Dict[str, Any], x: Tensor, alpha_condition: Optional[Tensor],
             rgb_condition: Optional[Tensor], cfg, name: Optional[str] = None) -> Tuple[Tensor, Tensor]:
  """modules.py:65-169.  x (B,S,P) -> rgb (B,S,3), alpha (B,S,1).

  Flax auto-names: MLP_0 = trunk, MLP_1 = rgb branch, MLP_2 = alpha branch
  (construction order modules.py:124-140).
  """
  global _SCOPE
  B, S, _ = x.shape
  x = x.reshape(B * S, -1)

  def bcast(c):
    return c[:, None, :].expand(B, S, c.shape[-1]).reshape(B * S, -1)

  sub = (lambda k: None) if name is None else (lambda k: f'{name}/{k}')
  prev, _SCOPE = _SCOPE, 'nerf_mlp'
  try:
    h = mlp(p['MLP_0'], x, cfg.nerf_trunk_depth, cfg.nerf_skips, False, sub('MLP_0'))
    if alpha_condition is not None or rgb_condition is not None:
      bottleneck = dense(p['bottleneck'], h)
    alpha_in = (torch.cat([bottleneck, bcast(alpha_condition)], -1)
                if alpha_condition is not None else h)
    alpha = mlp(p['MLP_2'], alpha_in, 0, (), True)
    rgb_in = (torch.cat([bottleneck, bcast(rgb_condition)], -1)
              if rgb_condition is not None else h)
    rgb = mlp(p['MLP_1'], rgb_in, cfg.nerf_rgb_branch_depth, (), True, sub('MLP_1'))
  finally:
    _SCOPE = prev
  return rgb.reshape(B, S, -1), alpha.reshape(B, S, -1)


# ----------------------------------------------------------------------------
# warping.py (SE3Field)
# ----------------------------------------------------------------------------
def se3_warp(p: Dict[str, Any], points: Tensor, metadata_embed: Tensor, alpha,
             num_warp_freqs: int, name: Optional[str] = None) -> Tensor:
  """warping.py:322-353 (use_pivot / use_translation off, the shipped default).

  points (...,3), metadata_embed (...,G).
  """
  points_embed = annealed_sinusoidal_encode(points, num_warp_freqs, alpha)
  inputs = torch.cat([points_embed, metadata_embed], -1)
  trunk = mlp(p['trunk'], inputs, len(p['trunk']), (4,), False, name)   # trunk_depth (warping.py:225): 6 unless warp_kwargs say otherwise
  w = dense(p['branches_w']['logit'], trunk)
  v = dense(p['branches_v']['logit'], trunk)
  theta = torch.linalg.norm(w, dim=-1)
  w = w / theta[..., None]
  v = v / theta[..., None]
  screw_axis = torch.cat([w, v], -1)
  transform = exp_se3(screw_axis, theta)
  warped = (transform @ to_homogenous(points)[..., None])[..., 0]
  return from_homogenous(warped)


def translation_warp(p: Dict[str, Any], points: Tensor, metadata_embed: Tensor, alpha,
                     num_warp_freqs: int, name: Optional[str] = None) -> Tensor:
  """TranslationField.warp (warping.py:155-164): points + MLP([annealed posenc, code]); the MLP is 6 x 128 with the
  skip at 4 and a 3-channel 'logit' output layer (warping.py:129-137)."""
  points_embed = annealed_sinusoidal_encode(points, num_warp_freqs, alpha)
  inputs = torch.cat([points_embed, metadata_embed], -1)
  return points + mlp(p['mlp'], inputs, len(p['mlp']) - 1, (4,), True, name)   # depth (warping.py:90): the hidden layers beside 'logit'


def time_encode(p, time, num_freqs, alpha=None):
  """modules.TimeEncoder (modules.py:297-322): MLP(depth 6, width 64, skips (4,), output_channels=features) on the
  annealed posenc of the time stamp; alpha=None -> num_freqs (full window).  time (...,1) -> (...,features)."""
  if alpha is None:
    alpha = num_freqs
  return mlp(p['mlp'], annealed_sinusoidal_encode(time, num_freqs, alpha), 6, (4,), True)


def se3_field(p, points, metadata, warp_alpha, num_warp_freqs,
              return_jacobian=False, metadata_encoded=False, name=None, time_alpha=None, num_time_freqs=1):
  """warping.py:355-389 (SE3Field) and :166-199 (TranslationField; selected by the parameter tree: 'mlp' instead of
  'trunk' + 'branches_*').  metadata: int ids (...,1)/(...), float time stamps (...,1) when the metadata encoder is the
  TimeEncoder (tree has metadata_encoder/mlp; warping.py:311-313), or encoded (...,G)."""
  se3_warp = globals()['translation_warp' if 'mlp' in p else 'se3_warp']
  if metadata_encoded:
    embed = metadata
  elif 'mlp' in p['metadata_encoder']:
    embed = time_encode(p['metadata_encoder'], metadata.to(points.dtype), num_time_freqs, time_alpha)
  else:
    ids = metadata[..., 0] if metadata.shape[-1] == 1 else metadata  # glo.py:50-53
    embed = p['metadata_encoder']['embed']['embedding'][ids.long()]
  out = {'warped_points': se3_warp(p, points, embed, warp_alpha, num_warp_freqs, name)}
  if return_jacobian:
    # jax.jacfwd(self.warp, argnums=0) per point (warping.py:385-387): column c
    # is the forward-mode derivative along e_c.
    flat_pts = points.reshape(-1, 3)
    flat_emb = embed.reshape(-1, embed.shape[-1])
    cols = []
    for c in range(3):
      tangent = torch.zeros_like(flat_pts)
      tangent[:, c] = 1.0
      _, jvp = torch.autograd.functional.jvp(
          lambda x: se3_warp(p, x, flat_emb, warp_alpha, num_warp_freqs, name),
          (flat_pts,), (tangent,), create_graph=torch.is_grad_enabled())
      cols.append(jvp)
    out['jacobian'] = torch.stack(cols, -1).reshape(*points.shape[:-1], 3, 3)
  return out


# ----------------------------------------------------------------------------
# model_utils.py
# ----------------------------------------------------------------------------
def sample_along_rays(origins, directions, num_coarse_samples, near, far,
                      use_stratified_sampling, use_linear_disparity,
                      t_rand: Optional[Tensor] = None):
  """model_utils.py:36-73.  t_rand (B,N) replaces random.uniform(key,...)."""
  B = origins.shape[0]
  dt = origins.dtype
  t_vals = torch.linspace(0., 1., num_coarse_samples, dtype=dt)
  if not use_linear_disparity:
    z_vals = near * (1. - t_vals) + far * t_vals
  else:
    z_vals = 1. / (1. / near * (1. - t_vals) + 1. / far * t_vals)
  if use_stratified_sampling:
    mids = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
    upper = torch.cat([mids, z_vals[..., -1:]], -1)
    lower = torch.cat([z_vals[..., :1], mids], -1)
    assert t_rand is not None
    z_vals = lower + (upper - lower) * t_rand
  else:
    z_vals = z_vals[None, :].expand(B, num_coarse_samples)
  pts = origins[..., None, :] + z_vals[..., :, None] * directions[..., None, :]
  return z_vals, pts


def compute_opaqueness_mask(weights, depth_threshold=0.5):
  """model_utils.py:218-240."""
  cum = torch.cumsum(weights, -1)
  opaq = cum >= depth_threshold
  padded = torch.cat([torch.zeros_like(opaq[..., :1]), opaq[..., :-1]], -1)
  return torch.logical_xor(opaq, padded).to(weights.dtype)


def compute_depth_index(weights, depth_threshold=0.5):
  """model_utils.py:243-246."""
  return torch.argmax(compute_opaqueness_mask(weights, depth_threshold), -1)


def compute_depth_map(weights, z_vals, depth_threshold=0.5):
  """model_utils.py:249-263."""
  return (compute_opaqueness_mask(weights, depth_threshold) * z_vals).sum(-1)


def volumetric_rendering(rgb, sigma, z_vals, dirs, use_white_background,
                         sample_at_infinity=True, eps=1e-10):
  """model_utils.py:76-136.  Always returns weights."""
  last = 1e10 if sample_at_infinity else 1e-19
  dists = torch.cat([z_vals[..., 1:] - z_vals[..., :-1],
                     torch.full_like(z_vals[..., :1], last)], -1)
  dists = dists * torch.linalg.norm(dirs[..., None, :], dim=-1)
  alpha = 1.0 - torch.exp(-sigma * dists)
  accum_prod = torch.cat([
      torch.ones_like(alpha[..., :1]),
      torch.cumprod(1.0 - alpha[..., :-1] + eps, -1)], -1)
  weights = alpha * accum_prod
  out_rgb = (weights[..., None] * rgb).sum(-2)
  exp_depth = (weights * z_vals).sum(-1)
  med_depth = compute_depth_map(weights, z_vals)
  acc = weights.sum(-1)
  if use_white_background:
    out_rgb = out_rgb + (1. - acc[..., None])
  if sample_at_infinity:
    acc = weights[..., :-1].sum(-1)
  return {'rgb': out_rgb, 'depth': exp_depth, 'med_depth': med_depth,
          'acc': acc, 'weights': weights}


def piecewise_constant_pdf(bins, weights, num_samples, use_stratified_sampling,
                           u: Optional[Tensor] = None):
  """model_utils.py:139-187 (literal masked max/min form).  u (B,N) replaces
  random.uniform when stratified."""
  eps = 1e-5
  weights = weights + eps
  pdf = weights / weights.sum(-1, keepdim=True)
  cdf = torch.cumsum(pdf, -1)
  cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
  if use_stratified_sampling:
    assert u is not None
  else:
    u = torch.linspace(0., 1., num_samples, dtype=bins.dtype)
    u = u.expand(*cdf.shape[:-1], num_samples)
  mask = u[..., None, :] >= cdf[..., :, None]

  def minmax(x):
    x0 = torch.where(mask, x[..., None], x[..., :1, None]).amax(-2)
    x1 = torch.where(~mask, x[..., None], x[..., -1:, None]).amin(-2)
    x0 = torch.minimum(x0, x[..., -2:-1])
    x1 = torch.maximum(x1, x[..., 1:2])
    return x0, x1

  bins_g0, bins_g1 = minmax(bins)
  cdf_g0, cdf_g1 = minmax(cdf)
  denom = cdf_g1 - cdf_g0
  denom = torch.where(denom < eps, torch.ones_like(denom), denom)
  t = (u - cdf_g0) / denom
  z_samples = bins_g0 + t * (bins_g1 - bins_g0)
  return z_samples.detach()  # lax.stop_gradient, model_utils.py:187


def sample_pdf(bins, weights, origins, directions, z_vals, num_samples,
               use_stratified_sampling, u=None):
  """model_utils.py:190-215."""
  z_samples = piecewise_constant_pdf(bins, weights, num_samples,
                                     use_stratified_sampling, u)
  z_vals = torch.sort(torch.cat([z_vals, z_samples], -1), -1).values
  pts = origins[..., None, :] + z_vals[..., None] * directions[..., None, :]
  return z_vals, pts


# ----------------------------------------------------------------------------
# utils.py
# ----------------------------------------------------------------------------
def general_loss_with_squared_residual(squared_x, alpha, scale):
  """utils.py:264-331: every branch (alpha = -inf, 0, 2, +inf and the generic one; the presets' call sites pass
  alpha=-2: training.py:112-113, :133-134; warp_reg_loss_alpha is a user-settable ScalarParam, training.py:38)."""
  eps = float(np.finfo(np.float32).eps)
  squared_scaled_x = squared_x / (scale ** 2)
  if alpha == -math.inf:
    loss = -torch.expm1(-0.5 * squared_scaled_x)                                   # utils.py:309
  elif alpha == 0:
    loss = torch.log1p(torch.clamp(0.5 * squared_scaled_x, max=3e37))              # :307, log1p_safe
  elif alpha == 2:
    loss = 0.5 * squared_scaled_x                                                  # :305
  elif alpha == math.inf:
    loss = torch.expm1(torch.clamp(0.5 * squared_scaled_x, max=87.5))              # :311, expm1_safe
  else:
    beta_safe = max(eps, abs(alpha - 2.))
    alpha_safe = (1.0 if alpha >= 0 else -1.0) * max(eps, abs(alpha))
    loss = (beta_safe / alpha_safe) * (
        torch.pow(squared_scaled_x / beta_safe + 1., 0.5 * alpha) - 1.)
  return scale * loss


def compute_psnr(mse):
  return -10. * torch.log(mse) / math.log(10.)  # utils.py:94-103


# ----------------------------------------------------------------------------
# models.py
# ----------------------------------------------------------------------------
class ModelSpec:
  """The NerfModel attributes that matter on the hot path (models.py:75-119),
  with the configs.ModelConfig defaults (configs.py:35-105)."""

  def __init__(self, **kw):
    self.num_coarse_samples = 64
    self.num_fine_samples = 128
    self.use_viewdirs = True
    self.near = 0.0206
    self.far = 0.826
    self.nerf_trunk_depth = 8
    self.nerf_trunk_width = 256
    self.nerf_rgb_branch_depth = 1
    self.nerf_rgb_branch_width = 128
    self.nerf_skips = (4,)
    self.use_stratified_sampling = False
    self.num_nerf_point_freqs = 8
    self.num_nerf_viewdir_freqs = 4
    self.num_appearance_embeddings = 4
    self.num_camera_embeddings = 2
    self.num_warp_embeddings = 4
    self.num_appearance_features = 8
    self.num_camera_features = 2
    self.num_warp_features = 8
    self.num_warp_freqs = 8
    self.sigma_activation = 'softplus'   # defaults.gin:66 (dataclass default relu)
    self.use_white_background = False
    self.use_linear_disparity = False
    self.use_sample_at_infinity = True
    self.use_appearance_metadata = False
    self.use_camera_metadata = False
    self.use_warp = False
    self.warp_field_type = 'se3'         # warp_defaults.gin; 'translation' = the ModelConfig dataclass default
    self.use_alpha_condition = False
    self.use_rgb_condition = False
    self.noise_std = None                # models.py:80; defaults.gin leaves it unset
    self.warp_metadata_encoder_type = 'glo'   # configs.py:101; 'time' = modules.TimeEncoder on metadata['time']
    self.num_time_encoder_freqs = 1      # metadata_encoder_num_freqs (warping.py:234)
    self.warp_trunk_depth = 6            # ModelConfig.warp_kwargs (configs.py:105): SE3Field trunk_depth / TranslationField depth
    self.warp_trunk_width = 128          # ... trunk_width / hidden_channels (warping.py:225-226, 90-91)
    for k, v in kw.items():
      if not hasattr(self, k):
        raise AttributeError(k)
      setattr(self, k, v)

  # derived widths
  @property
  def point_feat(self):
    return 3 + 6 * self.num_nerf_point_freqs

  @property
  def viewdir_feat(self):
    return 3 + 6 * self.num_nerf_viewdir_freqs

  @property
  def rgb_cond_width(self):
    w = self.viewdir_feat if self.use_viewdirs else 0
    if self.use_appearance_metadata and self.use_alpha_condition:
      w += self.num_appearance_features   # models.py:206 quirk
    if self.use_camera_metadata:
      w += self.num_camera_features
    return w

  @property
  def alpha_cond_width(self):
    return (self.num_appearance_features
            if self.use_appearance_metadata and self.use_alpha_condition else 0)

  @property
  def warp_in(self):
    return 3 + 6 * self.num_warp_freqs + self.num_warp_features


def _glorot(rng, fan_in, fan_out):
  lim = math.sqrt(6.0 / (fan_in + fan_out))
  return rng.uniform(-lim, lim, size=(fan_in, fan_out))


def init_params(spec: ModelSpec, seed=0, trained_like=False, dtype=torch.float64):
  """Parameter tree with the flax names of SURVEY.md A.2.  Init as the
  reference (glorot kernels, zero biases modules.py:107-108; embeddings U[0,.05)
  glo.py:33; warp heads U[0,1e-4) warping.py:238-239).  trained_like=True makes
  biases N(0,0.1) and scales the warp heads so theta is O(0.1): the reference
  init makes the warp ~identity and would hide bugs."""
  rng = np.random.default_rng(seed)

  def dense_p(i, o, kernel=None):
    k = _glorot(rng, i, o) if kernel is None else kernel
    b = rng.normal(0, 0.1, size=(o,)) if trained_like else np.zeros((o,))
    return {'kernel': torch.tensor(k, dtype=dtype), 'bias': torch.tensor(b, dtype=dtype)}

  def nerf_mlp_p():
    W, P = spec.nerf_trunk_width, spec.point_feat
    trunk = {}
    for i in range(spec.nerf_trunk_depth):
      fin = P if i == 0 else W
      if i in spec.nerf_skips:
        fin += P
      trunk[f'hidden_{i}'] = dense_p(fin, W)
    p = {'MLP_0': trunk}
    has_cond = spec.rgb_cond_width > 0 or spec.alpha_cond_width > 0
    if has_cond:
      p['bottleneck'] = dense_p(W, W)
    rgb = {}
    fin = (W + spec.rgb_cond_width) if spec.rgb_cond_width > 0 else W
    for i in range(spec.nerf_rgb_branch_depth):
      rgb[f'hidden_{i}'] = dense_p(fin, spec.nerf_rgb_branch_width)
      fin = spec.nerf_rgb_branch_width
    rgb['logit'] = dense_p(fin, 3)
    p['MLP_1'] = rgb
    afin = (W + spec.alpha_cond_width) if spec.alpha_cond_width > 0 else W
    p['MLP_2'] = {'logit': dense_p(afin, 1)}
    return p

  params = {'nerf_mlps_coarse': nerf_mlp_p()}
  if spec.num_fine_samples > 0:
    params['nerf_mlps_fine'] = nerf_mlp_p()
  if spec.use_warp:
    Ww = spec.warp_in
    trunk = {}
    WD, WW = spec.warp_trunk_depth, spec.warp_trunk_width
    for i in range(WD):
      fin = Ww if i == 0 else WW
      if i == 4:
        fin += Ww
      trunk[f'hidden_{i}'] = dense_p(fin, WW)
    head_scale = 0.3 if trained_like else 1e-4
    if spec.warp_metadata_encoder_type == 'time':   # modules.TimeEncoder: xavier hidden layers, uniform(0.05) output layer
      Tin = 1 + 2 * spec.num_time_encoder_freqs
      tm = {f'hidden_{i}': dense_p((Tin if i == 0 else 64) + (Tin if i == 4 else 0), 64) for i in range(6)}
      tm['logit'] = dense_p(64, spec.num_warp_features, rng.uniform(0, 0.05, size=(64, spec.num_warp_features)))
      meta_enc = {'mlp': tm}
    else:
      meta_enc = {'embed': {'embedding': torch.tensor(
          rng.uniform(0, 0.05, size=(spec.num_warp_embeddings, spec.num_warp_features)), dtype=dtype)}}
    if spec.warp_field_type == 'translation':     # warping.py:62-137: one MLP with a 3-channel output layer
      params['warp_field'] = {
          'metadata_encoder': meta_enc,
          'mlp': dict(trunk, logit=dense_p(WW, 3, rng.uniform(0, head_scale / 3, size=(WW, 3)))),
      }
    else:
      params['warp_field'] = {
          'metadata_encoder': meta_enc,
          'trunk': trunk,
          'branches_w': {'logit': dense_p(WW, 3, rng.uniform(0, head_scale, size=(WW, 3)))},
          'branches_v': {'logit': dense_p(WW, 3, rng.uniform(0, head_scale, size=(WW, 3)))},
      }
  if spec.use_appearance_metadata:
    params['appearance_encoder'] = {'embed': {'embedding': torch.tensor(
        rng.uniform(0, 0.05, size=(spec.num_appearance_embeddings, spec.num_appearance_features)), dtype=dtype)}}
  if spec.use_camera_metadata:
    params['camera_encoder'] = {'embed': {'embedding': torch.tensor(
        rng.uniform(0, 0.05, size=(spec.num_camera_embeddings, spec.num_camera_features)), dtype=dtype)}}
  return params


def tree_map(fn, tree):
  if isinstance(tree, dict):
    return {k: tree_map(fn, v) for k, v in tree.items()}
  return fn(tree)


def tree_leaves_with_path(tree, prefix=''):
  if isinstance(tree, dict):
    for k, v in tree.items():
      yield from tree_leaves_with_path(v, f'{prefix}/{k}' if prefix else k)
  else:
    yield prefix, tree


def _sigma_act(name, x):
  if name == 'softplus':
    return torch.nn.functional.softplus(x)
  if name == 'relu':
    return torch.relu(x)
  raise ValueError(name)


def get_condition_inputs(params, spec: ModelSpec, viewdirs, metadata,
                         metadata_encoded=False):
  """models.py:186-228, including the use_alpha_condition quirk at :206."""
  alpha_conditions, rgb_conditions = [], []
  if spec.use_viewdirs:
    rgb_conditions.append(sinusoidal_encode(viewdirs, spec.num_nerf_viewdir_freqs))
  if spec.use_appearance_metadata:
    if metadata_encoded:
      code = metadata['appearance']
    else:
      code = params['appearance_encoder']['embed']['embedding'][
          metadata['appearance'][..., 0].long()]
    if spec.use_alpha_condition:
      alpha_conditions.append(code)
    if spec.use_alpha_condition:   # sic (models.py:206)
      rgb_conditions.append(code)
  if spec.use_camera_metadata:
    if metadata_encoded:
      code = metadata['camera']
    else:
      code = params['camera_encoder']['embed']['embedding'][
          metadata['camera'][..., 0].long()]
    rgb_conditions.append(code)
  alpha_c = torch.cat(alpha_conditions, -1) if alpha_conditions else None
  rgb_c = torch.cat(rgb_conditions, -1) if rgb_conditions else None
  return alpha_c, rgb_c


def noise_regularize(raw_alpha, noise_std, use_stratified_sampling, noise):
  """model_utils.py:266-282: raw density += noise_std * N(0,1) (`noise` = the standard normals random.normal would
  draw, same shape as raw_alpha)."""
  if noise_std is not None and noise_std > 0.0 and use_stratified_sampling:
    assert noise is not None
    raw_alpha = raw_alpha + noise_std * noise
  return raw_alpha


def render_samples(params, spec: ModelSpec, level, points, z_vals, directions,
                   viewdirs, metadata, warp_alpha, use_warp, use_warp_jacobian,
                   metadata_encoded=False, return_points=False, noise=None, time_alpha=None):
  """models.py:230-287."""
  alpha_c, rgb_c = get_condition_inputs(params, spec, viewdirs, metadata,
                                        metadata_encoded)
  out = {}
  if return_points:
    out['points'] = points
  if use_warp:
    wmeta = metadata['time'] if spec.warp_metadata_encoder_type == 'time' else metadata['warp']   # models.py:252-254
    ch = spec.num_warp_features if metadata_encoded else 1
    wmeta = wmeta[:, None, :].expand(points.shape[0], points.shape[1], ch)
    wout = se3_field(params['warp_field'], points, wmeta, warp_alpha,
                     spec.num_warp_freqs, use_warp_jacobian, metadata_encoded, f'{level}/warp',
                     time_alpha, spec.num_time_encoder_freqs)
    points = wout['warped_points']
    if 'jacobian' in wout:
      out['warp_jacobian'] = wout['jacobian']
    if return_points:
      out['warped_points'] = wout['warped_points']
  points_embed = sinusoidal_encode(points, spec.num_nerf_point_freqs)
  raw_rgb, raw_alpha = nerf_mlp(params[f'nerf_mlps_{level}'], points_embed,
                                alpha_c, rgb_c, spec, level)
  raw_alpha = noise_regularize(raw_alpha, spec.noise_std, spec.use_stratified_sampling,
                               None if noise is None else noise[..., None])
  rgb = torch.sigmoid(raw_rgb)
  sigma = _sigma_act(spec.sigma_activation, raw_alpha[..., 0])
  out.update(volumetric_rendering(
      rgb, sigma, z_vals, directions, spec.use_white_background,
      spec.use_sample_at_infinity))
  return out


def nerf_model_apply(params, spec: ModelSpec, rays_dict, warp_alpha=0.0,
                     metadata_encoded=False, use_warp=True, return_points=False,
                     return_warp_jacobian=False, use_warp_jacobian=False,
                     t_rand=None, u=None, fixed_fine_z=None, noise_coarse=None, noise_fine=None, time_alpha=0.0):
  """models.py:289-375.  t_rand (B,N_c) / u (B,N_f) stand in for the 'coarse'
  and 'fine' RNG streams; always returns weights for both levels.
  fixed_fine_z (test hook): use these fine z_vals instead of resampling, so a
  finite-difference probe sees the same stop_gradient the autodiff does."""
  use_warp = spec.use_warp and use_warp
  origins, directions = rays_dict['origins'], rays_dict['directions']
  metadata = rays_dict.get('metadata', {})
  viewdirs = rays_dict.get('viewdirs', directions)
  z_vals, points = sample_along_rays(
      origins, directions, spec.num_coarse_samples, spec.near, spec.far,
      spec.use_stratified_sampling, spec.use_linear_disparity, t_rand)
  coarse = render_samples(
      params, spec, 'coarse', points, z_vals, directions, viewdirs, metadata,
      warp_alpha, use_warp, return_warp_jacobian or use_warp_jacobian,
      metadata_encoded, return_points, noise_coarse, time_alpha)
  coarse['z_vals'] = z_vals
  out = {'coarse': coarse}
  if spec.num_fine_samples > 0:
    z_mid = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
    z_f, points_f = sample_pdf(
        z_mid, coarse['weights'][..., 1:-1], origins, directions, z_vals,
        spec.num_fine_samples, spec.use_stratified_sampling, u)
    if fixed_fine_z is not None:
      z_f = fixed_fine_z
      points_f = origins[..., None, :] + z_f[..., None] * directions[..., None, :]
    fine = render_samples(
        params, spec, 'fine', points_f, z_f, directions, viewdirs, metadata,
        warp_alpha, use_warp, return_warp_jacobian, metadata_encoded,
        return_points, noise_fine, time_alpha)
    fine['z_vals'] = z_f
    out['fine'] = fine
  return out


# ----------------------------------------------------------------------------
# training.py
# ----------------------------------------------------------------------------
def jacobian_to_curl(jacobian):
  """utils.py:71-84."""
  return torch.stack([jacobian[..., 2, 1] - jacobian[..., 1, 2],
                      jacobian[..., 0, 2] - jacobian[..., 2, 0],
                      jacobian[..., 1, 0] - jacobian[..., 0, 1]], -1)


def jacobian_to_div(jacobian):
  """utils.py:87-91."""
  return torch.diagonal(jacobian, dim1=-2, dim2=-1).sum(-1) - 3.0


def compute_elastic_loss(jacobian, eps=1e-6, loss_type='log_svals'):
  """training.py:71-114 (every loss_type but 'nr', which the reference marks as producing NaNs, training.py:58)."""
  if loss_type == 'log_svals':
    svals = torch.linalg.svdvals(jacobian)
    log_svals = torch.log(torch.clamp(svals, min=eps))
    sq_residual = (log_svals ** 2).sum(-1)
  elif loss_type == 'svals':
    svals = torch.linalg.svdvals(jacobian)
    sq_residual = ((svals - 1.0) ** 2).sum(-1)
  elif loss_type == 'jtj':
    jtj = jacobian @ jacobian.transpose(-1, -2)
    sq_residual = ((jtj - torch.eye(3, dtype=jacobian.dtype)) ** 2).sum((-1, -2)) / 4.0
  elif loss_type == 'div':
    sq_residual = jacobian_to_div(jacobian) ** 2
  elif loss_type == 'det':
    sq_residual = (torch.linalg.det(jacobian) - 1.0) ** 2
  elif loss_type == 'log_det':
    sq_residual = torch.log(torch.clamp(torch.linalg.det(jacobian), min=eps)) ** 2
  else:
    raise NotImplementedError(loss_type)
  residual = torch.sqrt(sq_residual)
  loss = general_loss_with_squared_residual(sq_residual, alpha=-2.0, scale=0.03)
  return loss, residual


def compute_background_loss(params, spec, points, warp_ids_per_point, noise,
                            warp_alpha, alpha=-2, scale=0.001):
  """training.py:117-135 with the random ids / noise supplied by the caller."""
  points = points + noise
  wout = se3_field(params['warp_field'], points, warp_ids_per_point, warp_alpha,
                   spec.num_warp_freqs, False, False, 'background/warp')
  sq_residual = ((wout['warped_points'] - points) ** 2).sum(-1)
  return general_loss_with_squared_residual(sq_residual, alpha=alpha, scale=scale)


def loss_fn(params, spec: ModelSpec, batch, warp_alpha=0.0, t_rand=None, u=None,
            use_elastic_loss=False, elastic_loss_weight=0.0,
            elastic_reduce_method='weight', use_background_loss=False,
            background_loss_weight=0.0, background=None, fixed_fine_z=None,
            elastic_loss_type='log_svals', use_warp_reg_loss=False, warp_reg_loss_weight=0.0,
            warp_reg_loss_alpha=-2.0, warp_reg_loss_scale=0.001, noise_coarse=None, noise_fine=None, time_alpha=0.0):
  """training.py:171-262 (_compute_loss_and_stats + _loss_fn)."""
  ret = nerf_model_apply(params, spec, batch, warp_alpha, t_rand=t_rand, u=u,
                         use_warp_jacobian=use_elastic_loss,
                         fixed_fine_z=fixed_fine_z, return_points=use_warp_reg_loss,
                         noise_coarse=noise_coarse, noise_fine=noise_fine, time_alpha=time_alpha)
  total = 0.0
  stats = {}
  for level in ('fine', 'coarse'):
    if level not in ret:
      continue
    mo = ret[level]
    rgb_loss = ((mo['rgb'] - batch['rgb'][..., :3]) ** 2).mean()
    st = {'loss/rgb': rgb_loss}
    loss = rgb_loss
    if use_elastic_loss and level == 'coarse':
      weights = mo['weights'].detach()
      jac = mo['warp_jacobian']
      if elastic_reduce_method == 'median':
        idx = compute_depth_index(weights)
        jac = torch.take_along_dim(jac, idx[..., None, None, None], dim=-3)
      el, el_res = compute_elastic_loss(jac, loss_type=elastic_loss_type)
      if elastic_reduce_method == 'weight':
        el = weights * el
      el = el.sum(-1).mean()
      st['loss/elastic'] = el
      st['residual/elastic'] = el_res.mean()
      loss = loss + elastic_loss_weight * el
    if use_warp_reg_loss:   # training.py:199-212
      idx = compute_depth_index(mo['weights'].detach())
      warp_mag = ((mo['points'] - mo['warped_points']) ** 2).sum(-1)
      wr_res = torch.take_along_dim(warp_mag, idx[..., None], dim=-1)
      wr = general_loss_with_squared_residual(wr_res, alpha=warp_reg_loss_alpha, scale=warp_reg_loss_scale).mean()
      st['loss/warp_reg'] = wr
      st['residual/warp_reg'] = torch.sqrt(wr_res).mean()
      loss = loss + warp_reg_loss_weight * wr
    if 'warp_jacobian' in mo:   # training.py:214-222
      jac_all = mo['warp_jacobian']
      st['metric/jacobian_det'] = torch.linalg.det(jac_all).mean()
      st['metric/jacobian_div'] = jacobian_to_div(jac_all).mean()
      st['metric/jacobian_curl'] = torch.linalg.norm(jacobian_to_curl(jac_all), dim=-1).mean()
    st['loss/total'] = loss
    st['metric/psnr'] = compute_psnr(rgb_loss)
    stats[level] = st
    total = total + loss
  if use_background_loss:
    bg = compute_background_loss(params, spec, background['points'],
                                 background['warp_ids'], background['noise'],
                                 warp_alpha).mean()
    stats['background_loss'] = bg
    total = total + background_loss_weight * bg
  return total, stats, ret


def loss_and_grad(params, spec, batch, **kw):
  """jax.value_and_grad(_loss_fn) (training.py:264-265) via torch.autograd."""
  leaves = [(p, t) for p, t in tree_leaves_with_path(params)]
  req = [t.detach().clone().requires_grad_(True) for _, t in leaves]
  it = iter(req)
  params_r = tree_map(lambda _: next(it), params)
  total, stats, ret = loss_fn(params_r, spec, batch, **kw)
  grads = torch.autograd.grad(total, req, allow_unused=True)
  grads = [g if g is not None else torch.zeros_like(t) for g, t in zip(grads, req)]
  it = iter(grads)
  grad_tree = tree_map(lambda _: next(it), params)
  return total.detach(), stats, grad_tree, ret


def adam_update(p, m, v, g, step, lr, b1=0.9, b2=0.999, eps=1e-8):
  """flax.optim.Adam (flax 0.3.4 optim/adam.py; third-party, restated from its
  published rule).  `step` = number of updates applied before this
  one.  Returns (p, m, v)."""
  m = b1 * m + (1 - b1) * g
  v = b2 * v + (1 - b2) * g * g
  t = step + 1.0
  m_hat = m / (1 - b1 ** t)
  v_hat = v / (1 - b2 ** t)
  p = p - lr * m_hat / (torch.sqrt(v_hat) + eps)
  return p, m, v


# ----------------------------------------------------------------------------
# synthetic inputs
# ----------------------------------------------------------------------------
def synthetic_batch(num_rays, seed=0, dtype=torch.float64, num_ids=4,
                    num_camera_ids=2):
  rng = np.random.default_rng(seed)
  o = rng.uniform(-0.5, 0.5, size=(num_rays, 3))
  d = rng.normal(size=(num_rays, 3))
  d /= np.linalg.norm(d, axis=-1, keepdims=True)
  rgb = rng.uniform(0, 1, size=(num_rays, 3))
  ids = rng.integers(0, num_ids, size=(num_rays, 1))
  cam = rng.integers(0, num_camera_ids, size=(num_rays, 1))
  # metadata['time'] in [-1, 1] as datasets/core.py:272-274 builds it from the frame's time id
  time = ids.astype(np.float64) / max(num_ids - 1, 1) * 2.0 - 1.0
  return {
      'origins': torch.tensor(o, dtype=dtype),
      'directions': torch.tensor(d, dtype=dtype),
      'rgb': torch.tensor(rgb, dtype=dtype),
      'metadata': {'warp': torch.tensor(ids), 'appearance': torch.tensor(ids),
                   'camera': torch.tensor(cam), 'time': torch.tensor(time, dtype=dtype)},
  }
