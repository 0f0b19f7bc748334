"""Shared test plumbing: oracle ModelSpec -> nerfies_amd model on the GPU with identical parameters."""
import os
import types

import numpy as np
import torch

from oracle import nerfies_oracle as O

DEV = 'cuda:0'


def config_from_spec(spec):
  """An object with the configs.ModelConfig attribute names construct_nerf reads (configs.py:35-105)."""
  return types.SimpleNamespace(
      num_coarse_samples=spec.num_coarse_samples, num_fine_samples=spec.num_fine_samples,
      use_viewdirs=spec.use_viewdirs, nerf_trunk_depth=spec.nerf_trunk_depth, nerf_trunk_width=spec.nerf_trunk_width,
      nerf_rgb_branch_depth=spec.nerf_rgb_branch_depth, nerf_rgb_branch_width=spec.nerf_rgb_branch_width,
      nerf_skips=tuple(spec.nerf_skips), use_stratified_sampling=spec.use_stratified_sampling,
      num_nerf_point_freqs=spec.num_nerf_point_freqs, num_nerf_viewdir_freqs=spec.num_nerf_viewdir_freqs,
      sigma_activation=spec.sigma_activation, use_white_background=spec.use_white_background,
      use_linear_disparity=spec.use_linear_disparity, use_sample_at_infinity=spec.use_sample_at_infinity,
      use_appearance_metadata=spec.use_appearance_metadata, use_camera_metadata=spec.use_camera_metadata,
      appearance_metadata_dims=spec.num_appearance_features, camera_metadata_dims=spec.num_camera_features,
      use_warp=spec.use_warp, num_warp_freqs=spec.num_warp_freqs, num_warp_features=spec.num_warp_features,
      warp_field_type=spec.warp_field_type, use_alpha_condition=spec.use_alpha_condition, use_rgb_condition=spec.use_rgb_condition,
      noise_std=spec.noise_std, warp_metadata_encoder_type=spec.warp_metadata_encoder_type, warp_kwargs=warp_kwargs_from_spec(spec))


def warp_kwargs_from_spec(spec):
  """ModelConfig.warp_kwargs (configs.py:105) as a Gin file would bind them: the trunk's depth / width under the field's own attribute
  names (SE3Field trunk_depth / trunk_width, warping.py:225-226; TranslationField depth / hidden_channels, warping.py:90-91)."""
  d, w = getattr(spec, 'warp_trunk_depth', 6), getattr(spec, 'warp_trunk_width', 128)
  if not spec.use_warp or (d, w) == (6, 128):
    return {}
  dk, wk = ('depth', 'hidden_channels') if spec.warp_field_type == 'translation' else ('trunk_depth', 'trunk_width')
  return {dk: d, wk: w}


def gpu_model(spec, oparams, batch_size=0):
  """(model, FlatParams on DEV) holding exactly the oracle parameter tree `oparams`."""
  from nerfies_amd import models, params as P
  model, fp = models.construct_nerf(
      0, config_from_spec(spec), batch_size, list(range(spec.num_appearance_embeddings)),
      list(range(spec.num_camera_embeddings)), list(range(spec.num_warp_embeddings)), spec.near, spec.far)
  P.flat_from_tree(O.tree_map(lambda t: t.float(), oparams), model.layout, DEV, out=fp.flat)
  return model, fp


def gpu_batch(batch):
  out = {k: v.to(DEV).float() for k, v in batch.items() if torch.is_tensor(v)}
  out['metadata'] = {k: (v.to(DEV).float() if v.is_floating_point() else v.to(DEV)) for k, v in batch.get('metadata', {}).items()}
  return out


def flat_grad_from_tree(grads, layout):
  from nerfies_amd import params as P
  return P.flat_from_tree(O.tree_map(lambda t: t.float(), grads), layout, 'cpu')


# ---------------------------------------------------------------------------------------------
# ReLU sign bits of the last stashed forward (training workspace), decoded to boolean masks the
# oracle can be pinned to (oracle.relu_hook).  Layout: include/nerfies_amd.h nrf_debug_ws_offset.
# ---------------------------------------------------------------------------------------------
def _ws_words(model, ws, name, level, nwords):
  import ctypes as C
  from nerfies_amd import lib as L
  off = C.c_int64(0)
  L.check(model.lib.nrf_debug_ws_offset(model.handle, name.encode(), level, C.byref(off)), model.lib)
  return ws[off.value:off.value + nwords].view(torch.int32).cpu().numpy().view('uint32')


def _decode_bits(words, nlayers, ntiles, ncb, rows):
  """words: uint32 [nlayers][ntiles][4 waves][64 lanes][ncb] -> bool [nlayers][rows][4 * 32 * ncb]."""
  import numpy as np
  w = words.reshape(nlayers, ntiles, 4, 64, ncb)
  q, e = np.arange(8), np.arange(4)
  bits = ((w[..., None, None] >> (4 * q[:, None] + e[None, :]).astype('uint32')) & 1).astype(bool)   # [L][t][wave][lane][cb][q][e]
  lane = np.arange(64)
  j, h = lane & 31, lane >> 5
  g = (q[None, :] & 1) + 2 * h[:, None] + 4 * (q[None, :] >> 1)          # [lane][q]
  p = 4 * g[:, :, None] + e[None, None, :]                               # [lane][q][e] tile row
  wave, cb = np.arange(4), np.arange(ncb)
  n = wave[:, None, None] * 32 * ncb + 32 * cb[None, None, :] + j[None, :, None]   # [wave][lane][cb] feature
  out = np.zeros((nlayers, ntiles * 64, 4 * 32 * ncb), bool)
  t = np.arange(ntiles)
  rows_idx = (t[:, None, None, None, None, None] * 64 + p[None, None, :, None, :, :])          # [t][1][lane][1][q][e]
  rows_idx = np.broadcast_to(rows_idx, (ntiles, 4, 64, ncb, 8, 4))
  cols_idx = np.broadcast_to(n[None, :, :, :, None, None], (ntiles, 4, 64, ncb, 8, 4))
  for l in range(nlayers):
    out[l][rows_idx, cols_idx] = bits[l]
  return torch.from_numpy(out[:, :rows])


def trunk_layer_map(spec):
  """The chain layer (0..7) each of the caller's trunk layers runs on (csrc/nrf_api.hip nrf_create): in place, unless the trunk's skip
  s <= 4 with depth - s <= 4 is laid out around the kernels' own skip layer 4 (identity layers s..3 in between); the layers behind
  the caller's last one are identities as well and appear in no oracle hook."""
  xd = spec.nerf_trunk_depth
  xs = spec.nerf_skips[0] if (len(spec.nerf_skips) and spec.nerf_skips[0] < xd) else -1
  imap = list(range(8))
  if 0 <= xs <= 4 and xd - xs <= 4:
    for i in range(xs, xd):
      imap[i] = 4 + (i - xs)
  return imap[:xd] + [l for l in range(8) if l not in imap[:xd]]


def gpu_relu_masks(model, spec, num_rays, nbg=0, elastic=False):
  """{oracle hook name: [bool (rows, width) per layer]} read back from the training workspace of the last
  loss_and_grad / apply(train=True) call with these sizes."""
  ws = model.workspace(num_rays, True, DEV, nbg, elastic)
  torch.cuda.synchronize()
  masks = {}
  WW = getattr(spec, 'warp_trunk_width', 128)
  levels = [('coarse', 0, num_rays * spec.num_coarse_samples)]
  if spec.num_fine_samples > 0:
    levels.append(('fine', 1, num_rays * (spec.num_coarse_samples + spec.num_fine_samples)))
  for name, lv, rows in levels:
    nt = (rows + 63) // 64
    m = _decode_bits(_ws_words(model, ws, 'bits_trunk', lv, nt * 4 * 128 * 8), 8, nt, 2, rows)
    masks[f'{name}/MLP_0'] = [m[l][:, :spec.nerf_trunk_width] for l in trunk_layer_map(spec)]
    m = _decode_bits(_ws_words(model, ws, 'bits_rgbh', lv, nt * 4 * 64), 1, nt, 1, rows)
    masks[f'{name}/MLP_1'] = [m[0][:, :spec.nerf_rgb_branch_width]]
    if spec.use_warp:
      m = _decode_bits(_ws_words(model, ws, 'w_bits', lv, nt * 4 * 64 * 6), 6, nt, 1, rows)
      masks[f'{name}/warp'] = [m[l][:, :WW] for l in range(6)]
  if spec.use_warp and nbg > 0:
    nt = (nbg + 63) // 64
    m = _decode_bits(_ws_words(model, ws, 'w_bits', 2, nt * 4 * 64 * 6), 6, nt, 1, nbg)
    masks['background/warp'] = [m[l][:, :WW] for l in range(6)]
  return masks


class PinnedRelu:
  """oracle.relu_hook callable: activation = pre * mask with the HIP path's masks.  Records where the oracle's own sign
  disagrees: `flips` / `total` units and, per disagreeing unit, |pre| relative to its layer's rms (`mags`)."""

  def __init__(self, masks):
    self.masks = masks
    self.flips, self.total, self.mags = 0, 0, []

  def __call__(self, name, layer, pre):
    m = self.masks[name][layer].reshape(pre.shape)
    with torch.no_grad():
      bad = (pre.detach() > 0) != m
      nb = int(bad.sum())
      self.flips += nb
      self.total += m.numel()
      if nb:
        rms = float(pre.detach().pow(2).mean().sqrt())
        self.mags.append(pre.detach()[bad].abs().double() / max(rms, 1e-30))
    return pre * m.to(pre.dtype)

  @property
  def worst(self):
    return float(torch.cat(self.mags).max()) if self.mags else 0.0

  def quantile(self, q):
    return float(torch.cat(self.mags).quantile(q)) if self.mags else 0.0


# ---------------------------------------------------------------------------------------------
# Pinned-branch gradient parity (see tests/test_gpu_pinned.py for the rationale)
# ---------------------------------------------------------------------------------------------
FLIP_FRACTION = 2e-3     # units whose fp64 sign differs from the HIP path's
FLIP_PRE = 5e-3          # ... and the 95th percentile of their |pre| relative to the layer rms (rounding-level ties only)


def grad_tol(spec):
  """Per-leaf gradient tolerance, relative to the leaf's max-abs entry: 2e-3, and 4e-3 with the warp on at F_p = 9, 10.
  With the warp on, float32 rounding of the warped point (3e-8 absolute) is a phase error of 2^(F_p-1) * 3e-8 in the top
  posenc band; the per-sample dL/dx' (dominated by the top bands, magnitude ~2^F_p) cancels across samples into a much
  smaller parameter gradient, so the RELATIVE error of the warp leaves grows like 2^F_p in ANY float32 evaluation: the
  oracle's own float32 restatement, with identical branches and samples, is 4e-4 from float64 at F_p = 8 and 1-2e-3 at
  F_p = 10 (tests/test_pinned_host.py::test_float32_floor_of_the_warp_gradients)."""
  return 4e-3 if (spec.use_warp and spec.num_nerf_point_freqs > 8) else 2e-3


def host_threads(n):
  class _T:
    def __enter__(self):
      self.prev = torch.get_num_threads()
      torch.set_num_threads(max(1, min(n, os.cpu_count() or 1)))

    def __exit__(self, *a):
      torch.set_num_threads(self.prev)
  return _T()


def leaf(tree, path):
  node = tree
  for k in path.split('/'):
    node = node[k]
  return node


def run_pinned(spec, B, alpha, seed=3, strat=True, elastic=None, background=None, params=None, batch=None, t_rand=None, u=None,
               warp_reg=None, time_alpha=None):
  """GPU loss_and_grad, masks read back, fp64 oracle pinned to them.  Returns a dict of everything compared."""
  from nerfies_amd import params as P
  p64 = params if params is not None else O.init_params(spec, seed=seed, trained_like=True, dtype=torch.float64)
  b64 = batch if batch is not None else O.synthetic_batch(B, seed=seed + 1, dtype=torch.float64)
  model, fp = gpu_model(spec, p64, B)
  gb = gpu_batch(b64)
  rngs = None
  if spec.use_stratified_sampling and t_rand is None:
    g = torch.Generator().manual_seed(seed)
    t_rand = torch.rand(B, spec.num_coarse_samples, generator=g).double()
    u = torch.rand(B, spec.num_fine_samples, generator=g).double()
  if t_rand is not None:
    rngs = {'coarse': t_rand.float().to(DEV), 'fine': u.float().to(DEV)}
  gkw, okw, nbg = {}, {}, 0
  if background is not None:   # dict(points, warp_ids, noise, weight)
    nbg = background['points'].shape[0]
    gkw['background'] = {'points': (background['points'] + background['noise']).float().to(DEV),
                         'warp_ids': background['warp_ids'].to(DEV), 'weight': background['weight']}
    okw.update(use_background_loss=True, background_loss_weight=background['weight'],
               background={k: background[k] for k in ('points', 'warp_ids', 'noise')})
  if elastic is not None:      # dict(weight, reduce_method[, loss_type])
    gkw['elastic'] = dict(elastic)
    okw.update(use_elastic_loss=True, elastic_loss_weight=elastic['weight'], elastic_reduce_method=elastic.get('reduce_method', 'weight'),
               elastic_loss_type=elastic.get('loss_type', 'log_svals'))
  if warp_reg is not None:     # dict(weight[, alpha, scale])
    gkw['warp_reg'] = dict(warp_reg)
    okw.update(use_warp_reg_loss=True, warp_reg_loss_weight=warp_reg['weight'], warp_reg_loss_alpha=warp_reg.get('alpha', -2.0),
               warp_reg_loss_scale=warp_reg.get('scale', 0.001))
  if spec.noise_std and spec.use_stratified_sampling:   # explicit normals for model_utils.noise_regularize
    g2 = torch.Generator().manual_seed(seed + 77)
    nz_c = torch.randn(B, spec.num_coarse_samples, generator=g2).double()
    nz_f = torch.randn(B, spec.num_coarse_samples + spec.num_fine_samples, generator=g2).double()
    rngs = dict(rngs or {}, noise_coarse=nz_c.float().to(DEV), noise_fine=nz_f.float().to(DEV))
    okw.update(noise_coarse=nz_c, noise_fine=nz_f)
  if time_alpha is not None:   # warp_extra['time_alpha'] (TimeEncoder window, models.py:252-254)
    okw['time_alpha'] = time_alpha
  grad, stats = model.loss_and_grad(fp, gb, warp_extra={'alpha': alpha, 'time_alpha': time_alpha or 0.0}, rngs=rngs, **gkw)
  torch.cuda.synchronize()
  masks = gpu_relu_masks(model, spec, B, nbg, elastic is not None)
  hook = PinnedRelu(masks)
  # the fine samples are a stop_gradient input of the fine pass (model_utils.py:187): the oracle takes the HIP path's own
  # (they agree with its natural ones to ~1e-5, asserted below; through the 2^(F_p-1) posenc that difference alone
  # would move the gradients by more than the kernels' own error)
  z_fine = None
  if spec.num_fine_samples > 0:
    ws = model.workspace(B, True, DEV, nbg, elastic is not None)
    S1 = spec.num_coarse_samples + spec.num_fine_samples
    z_fine = _ws_words(model, ws, 'z', 1, B * S1).view('float32').reshape(B, S1)
    z_fine = torch.from_numpy(z_fine.copy()).double()
  with O.relu_hook(hook):
    loss, ostats, ograds, ret = O.loss_and_grad(p64, spec, b64, warp_alpha=alpha, t_rand=t_rand, u=u, fixed_fine_z=z_fine, **okw)
  if z_fine is not None:   # ... and its own resampling lands on the same depths
    with torch.no_grad():
      z_nat = O.sample_pdf(.5 * (ret['coarse']['z_vals'][..., 1:] + ret['coarse']['z_vals'][..., :-1]), ret['coarse']['weights'][..., 1:-1],
                           b64['origins'], b64['directions'], ret['coarse']['z_vals'], spec.num_fine_samples,
                           spec.use_stratified_sampling, u)[0]
    assert (z_nat - z_fine).abs().max().item() < 2e-4 * (spec.far - spec.near), (z_nat - z_fine).abs().max().item()
    assert (z_nat - z_fine).abs().mean().item() < 2e-6 * (spec.far - spec.near)
  got = P.tree_from_flat(grad.cpu(), model.layout)
  errs = {}
  for path, og in O.tree_leaves_with_path(ograds):
    scale = max(og.abs().max().item(), 1e-30)
    errs[path] = ((leaf(got, path).double() - og).abs().max().item() / scale, scale)
  return dict(model=model, fp=fp, gb=gb, rngs=rngs, stats=stats.cpu(), loss=loss.item(), ostats=ostats, ret=ret, errs=errs, hook=hook,
              alpha=alpha, time_alpha=time_alpha, tol=grad_tol(spec), spec=spec, p64=p64, b64=b64, t_rand=t_rand, u=u, okw=okw, got=got)


def assert_unpinned(r, label, factor=3.0):
  """The UNPINNED full-shape statistic: the same HIP gradient against the FREE-RUNNING float64 oracle -- its own
  ReLU branches, its own fine samples, nothing taken from the HIP path -- as per-leaf relative L2 distance, next to the distance of
  the oracle's own float32 evaluation from that float64 run.  float32 ties flip whole ReLU columns in ANY float32 evaluation
  order, so element-wise agreement is not a property a float32 path can have (that is what the pinned comparison removes); what
  can be asserted without conditioning on the HIP path's outputs is that it sits no further from float64 than `factor` x the
  reference arithmetic restated in float32 does."""
  spec, p64, b64 = r['spec'], r['p64'], r['b64']
  to32 = lambda t: t.float() if torch.is_tensor(t) and t.is_floating_point() else t
  _, _, g64, _ = O.loss_and_grad(p64, spec, b64, warp_alpha=r['alpha'], t_rand=r['t_rand'], u=r['u'], **r['okw'])
  p32 = O.tree_map(to32, p64)
  b32 = {k: (O.tree_map(to32, v) if isinstance(v, dict) else to32(v)) for k, v in b64.items()}
  okw32 = {k: (O.tree_map(to32, v) if isinstance(v, dict) else to32(v)) for k, v in r['okw'].items()}
  _, _, g32, _ = O.loss_and_grad(p32, spec, b32, warp_alpha=r['alpha'], t_rand=to32(r['t_rand']) if r['t_rand'] is not None else None,
                                 u=to32(r['u']) if r['u'] is not None else None, **okw32)
  worst = ('', 0.0, 0.0)
  rows = []
  for (path, want), (_, f32) in zip(O.tree_leaves_with_path(g64), O.tree_leaves_with_path(g32)):
    nrm = max(want.norm().item(), 1e-30)
    l2_gpu = (leaf(r['got'], path).double() - want).norm().item() / nrm
    l2_f32 = (f32.double() - want).norm().item() / nrm
    rows.append((path, l2_gpu, l2_f32))
    if l2_gpu > worst[1]:
      worst = (path, l2_gpu, l2_f32)
  med_gpu = sorted(x[1] for x in rows)[len(rows) // 2]
  med_f32 = sorted(x[2] for x in rows)[len(rows) // 2]
  print(f'[{label}, UNPINNED vs the free-running float64 oracle] per-leaf relative L2: worst {worst[0]} hip {worst[1]:.2e} '
        f'(float32 oracle {worst[2]:.2e}); median hip {med_gpu:.2e} / float32 oracle {med_f32:.2e}')
  for path, l2_gpu, l2_f32 in rows:
    assert l2_gpu <= factor * l2_f32 + 1e-5, (label, path, l2_gpu, l2_f32)


def assert_pinned(r, label, loss_tol=1e-5):
  assert abs(r['stats'][4].item() - r['loss']) < loss_tol, (label, r['stats'][4].item(), r['loss'])
  worst = max(r['errs'].items(), key=lambda kv: kv[1][0])
  print(f'[{label}] loss gpu {r["stats"][4].item():.7f} oracle {r["loss"]:.7f}; worst leaf {worst[0]} rel err {worst[1][0]:.2e}; '
        f'ReLU ties {r["hook"].flips}/{r["hook"].total} (|pre|/rms: 95 % below {r["hook"].quantile(0.95):.1e}, max {r["hook"].worst:.1e})')
  for path, (err, scale) in r['errs'].items():
    assert err < r['tol'], (label, path, err, scale)
  h = r['hook']
  assert h.flips <= FLIP_FRACTION * h.total, (label, h.flips, h.total)
  assert h.quantile(0.95) < FLIP_PRE, (label, h.quantile(0.95), h.worst)


def assert_forward(r, spec, atol=1e-4):
  """The rendered outputs of a (non-training) forward on the same rays against the pinned oracle's."""
  out = r['model'].apply({'params': r['fp']}, r['gb'], {'alpha': r['alpha'], 'time_alpha': r.get('time_alpha') or 0.0}, rngs=r['rngs'],
                         return_weights=True)
  for lv in out:
    for k in ('rgb', 'depth', 'acc', 'weights'):
      np.testing.assert_allclose(out[lv][k].cpu().numpy(), r['ret'][lv][k].detach().numpy(), atol=atol, err_msg=f'{lv}/{k}')




# ---------------------------------------------------------------------------------------------
# bf16 training stash (csrc/nrf_internal.h BfStash) decoded to [rows, features] float64 matrices
# ---------------------------------------------------------------------------------------------
def bf16_stash(model, ws, name, level, nlayers, nblocks, rows, as_float32=False, ngroups=None):
  """Buffer `name` ("b_pe", "b_h", "b_bn", "b_rgbh", "b_dy", "b_dbn", "b_drgbh", "b_dsmall") of `level`: [nlayers] matrices
  (rows, 32 * nblocks).  Layout per 32-sample group: [block b][jp][lane = n + 32 h][8 bf16], the 8 = features
  32 b + 8 (2 jp + jj) + 4 h + i in (jj, i) order."""
  ng = ngroups if ngroups is not None else (rows + 255) // 256 * 8
  words = _ws_words(model, ws, name, level, nlayers * ng * nblocks * 512)
  a = words.view('uint16').reshape(nlayers, ng, nblocks, 2, 2, 32, 2, 4)        # [L][g][b][jp][h][n][jj][i]
  a = a.transpose(0, 1, 5, 2, 3, 6, 4, 7).reshape(nlayers, ng * 32, nblocks * 32)  # [L][g, n][b, jp, jj, h, i]
  f = (a.astype('uint32') << 16).view('float32')
  if not as_float32:   # float32 holds a bfloat16 exactly: the full-shape tests keep it to halve the host memory
    f = f.astype('float64')
  return [torch.from_numpy(f[l, :rows].copy()) for l in range(nlayers)]


def bf16_round(t):
  return t.float().to(torch.bfloat16).double()


# ---------------------------------------------------------------------------------------------
# tests/golden/make_reference_vectors.py::loss_directional -- the seeded parameter directions and the cases, shared by the CPU
# (oracle autograd vs the reference's finite difference) and the GPU test (HIP gradient vs the same numbers)
# ---------------------------------------------------------------------------------------------
LOSS_DIR_CASES = {
    'nowarp': dict(spec=dict(num_coarse_samples=64, num_fine_samples=128, num_nerf_point_freqs=8, use_stratified_sampling=True),
                   B=12, alpha=0.0, bg=0),
    'warp_bg': dict(spec=dict(num_coarse_samples=24, num_fine_samples=24, num_nerf_point_freqs=6, use_stratified_sampling=True, use_warp=True,
                              num_warp_freqs=5, num_warp_features=8, use_camera_metadata=True), B=8, alpha=3.25, bg=9),
}


def loss_directions(params, seed, ndir):
  """[{path tuple: float64 array}] exactly as make_reference_vectors.loss_directions draws them: leaves in sorted path order,
  standard normals scaled by the leaf's rms + 1e-3."""
  leaves = []

  def walk(t, path):
    for k in sorted(t):
      if isinstance(t[k], dict):
        walk(t[k], path + (k,))
      else:
        leaves.append((path + (k,), np.asarray(t[k].detach().double().numpy() if torch.is_tensor(t[k]) else t[k], dtype=np.float64)))
  walk(params, ())
  rng = np.random.default_rng(seed)
  dirs = []
  for _ in range(ndir):
    dirs.append({path: rng.standard_normal(a.shape) * (np.sqrt(np.mean(a * a)) + 1e-3) for path, a in leaves})
  return dirs


def tree_dot(grads, direction, path=()):
  """<grad tree, direction> in float64."""
  tot = 0.0
  for k, v in grads.items():
    if isinstance(v, dict):
      tot += tree_dot(v, direction, path + (k,))
    else:
      g = v.detach().double().cpu().numpy() if torch.is_tensor(v) else np.asarray(v, dtype=np.float64)
      tot += float((g * direction[path + (k,)]).sum())
  return tot

