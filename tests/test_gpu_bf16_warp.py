"""SE3 trunk on bfloat16 operands (csrc/warp_bf16.hip; NRF_FLAG_BF16 with the warp field, BASELINE configs[3]).

The reference has no reduced-precision mode (warping.py:264-288 runs in float32), so parity is established as for the NeRF MLPs
(tests/test_gpu_bf16_train.py):
  1. GIVEN THE KERNELS' OWN STASH, in float64 with the same roundings: every stashed quantity of every pass through the field
     (coarse / fine samples, background points, the three tangents per coarse sample) is recomputed from the stashed quantity in
     front of it -- trunk input from the points, h_{l+1} from h_l, the head outputs (w, v) from h_6, dpre_{l-1} from dpre_l and
     the ReLU mask, the warp leaves' gradients from the (X, dY) stashes summed over the passes.  This pins layouts, masks, weight
     streams, the panel pipeline and the wgrad plumbing without the tie amplification an end-to-end comparison suffers.
  2. AGAINST THE float32 TRUNK on identical rays (bf16='mlp' keeps it in float32, everything else equal): warped points, loss,
     elastic / background loss values (within 2 %), every gradient leaf's direction (cos >= 0.97).
The end-to-end training gate (held-out PSNR with the warp on) is tests/test_gpu_bf16_convergence.py."""
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
ALPHA = 4.3
q = H.bf16_round


def _setup(B, seed=5, nbg=0, head_scale=1.0, **kw):
  spec = O.ModelSpec(**dict(dict(num_coarse_samples=32, num_fine_samples=32, num_nerf_point_freqs=8, use_stratified_sampling=True,
                                 use_warp=True, num_warp_freqs=8, num_warp_features=8), **kw))
  p = O.init_params(spec, seed=seed, trained_like=True, dtype=torch.float64)
  if head_scale != 1.0:   # the "trained-like" heads throw points several scene sizes away; a capture's deformations are ~ 0.05
    heads = [p['warp_field']['mlp']] if spec.warp_field_type == 'translation' else [p['warp_field'][br] for br in ('branches_w', 'branches_v')]
    for hd in heads:
      for k in ('kernel', 'bias'):
        hd['logit'][k] = hd['logit'][k] * head_scale
  b = O.synthetic_batch(B, seed=seed + 1, dtype=torch.float64)
  g = torch.Generator().manual_seed(seed + 2)
  rngs = {'coarse': torch.rand(B, spec.num_coarse_samples, generator=g).to(DEV),
          'fine': torch.rand(B, spec.num_fine_samples, generator=g).to(DEV)}
  model, fp = H.gpu_model(spec, p, B)
  bg = None
  if nbg:
    bg = {'points': ((torch.rand(nbg, 3, generator=g) - 0.5) * 0.8).to(DEV), 'warp_ids': torch.randint(0, 4, (nbg,), generator=g).to(DEV),
          'weight': 1.0}
  return spec, p, b, model, fp, rngs, bg


def _f32rows(model, ws, name, lv, rows, width):
  return torch.from_numpy(H._ws_words(model, ws, name, lv, rows * width).view('float32').reshape(rows, width).copy()).double()


def _window(alpha, F):
  k = torch.arange(F, dtype=torch.float64)
  return 0.5 * (1 + torch.cos(np.pi * torch.clip(torch.tensor(alpha, dtype=torch.float64) - k, 0, 1) + np.pi))


def _trunk_input(x, code, alpha, F, tangent_dir=None):
  """[annealed posenc(x), code] (warping.py:326-327, modules.py:231-294) or its derivative along one coordinate."""
  w = _window(alpha, F)
  cols = []
  if tangent_dir is None:
    cols.append(x)
    for f in range(F):
      a = (x.float() * float(2 ** f)).double()
      cols += [w[f] * torch.sin(a), w[f] * torch.sin((a.float() + np.float32(np.pi / 2)).double())]
    cols.append(code)
  else:
    e = torch.zeros_like(x); e[:, tangent_dir] = 1
    cols.append(e)
    for f in range(F):
      a = (x.float() * float(2 ** f)).double()
      cols += [w[f] * 2.0 ** f * torch.cos(a) * e, -w[f] * 2.0 ** f * torch.sin(a) * e]
    cols.append(torch.zeros_like(code))
  return torch.cat(cols, -1)


def _close(got, want, what, atol_frac=2e-2, l2=5e-3, ulp_frac=0.05):
  got, want = got.double(), want.double()
  scale = want.abs().max().item()
  err = (got - want).abs()
  assert err.max().item() <= atol_frac * scale + 1e-30, (what, err.max().item(), scale)
  if scale > 0:
    assert (err.norm() / want.norm()).item() <= l2, (what, (err.norm() / want.norm()).item())
    frac = (err > 2.0 ** -7 * want.abs() + 1e-6 * scale).double().mean().item()
    assert frac < ulp_frac, (what, frac)


def test_bf16_warp_stash_chain_and_leaf_gradients_given_the_stash():
  from nerfies_amd import params as P
  B, nbg = 21, 100
  spec, p, b, model, fp, rngs, bg = _setup(B, nbg=nbg, use_camera_metadata=True)
  F, G = spec.num_warp_freqs, spec.num_warp_features
  Win = 3 + 6 * F + G
  gb = H.gpu_batch(b)
  grad, stats = model.loss_and_grad(fp, gb, warp_extra={'alpha': ALPHA}, rngs=rngs, bf16=True, background=bg,
                                    elastic={'weight': 0.01, 'reduce_method': 'weight'})
  torch.cuda.synchronize()
  assert torch.isfinite(grad).all() and torch.isfinite(stats).all()
  ws = model.workspace(B, True, DEV, nbg, True, bf16=True)
  got = P.tree_from_flat(grad.cpu(), model.layout)['warp_field']
  wf = O.tree_map(lambda t: t.float().double(), p['warp_field'])
  Wk = [wf['trunk'][f'hidden_{l}']['kernel'] for l in range(6)]
  bk = [wf['trunk'][f'hidden_{l}']['bias'] for l in range(6)]
  Wh = torch.cat([wf['branches_w']['logit']['kernel'], wf['branches_v']['logit']['kernel']], 1)   # [128, 6]
  bh = torch.cat([wf['branches_w']['logit']['bias'], wf['branches_v']['logit']['bias']])
  table = wf['metadata_encoder']['embed']['embedding']
  hilo = lambda v: q(v) + q(v - q(v))            # biases ride as a (hi, lo) bfloat16 pair
  S = (spec.num_coarse_samples, spec.num_coarse_samples + spec.num_fine_samples)
  want = {f'trunk/hidden_{l}/{k}': 0.0 for l in range(6) for k in ('kernel', 'bias')}
  want.update({'branches_w/logit/kernel': 0.0, 'branches_w/logit/bias': 0.0, 'branches_v/logit/kernel': 0.0, 'branches_v/logit/bias': 0.0})
  prim_h0 = None
  worst = 0.0
  # (pass name, level in the workspace, rows, tangent?)
  for name, lv, rows, tangent in (('coarse', 0, B * S[0], False), ('fine', 1, B * S[1], False), ('background', 2, nbg, False),
                                  ('tangent', 3, B * S[0], True)):
    ngp = (rows + 255) // 256 * 8                      # groups of the (primal) level
    ng = 3 * ngp if tangent else ngp
    nrow = ng * 32
    X = H.bf16_stash(model, ws, 'bw_in', lv, 1, 2, nrow, ngroups=ng)[0]
    Hs = H.bf16_stash(model, ws, 'bw_h', lv, 6, 4, nrow, ngroups=ng)
    DY = H.bf16_stash(model, ws, 'bw_dy', lv, 6, 4, nrow, ngroups=ng)
    DH = H.bf16_stash(model, ws, 'bw_dhead', lv, 1, 2, nrow, ngroups=ng)[0]
    valid = torch.zeros(nrow, dtype=torch.bool)
    if tangent:
      for c in range(3):
        valid[c * ngp * 32:c * ngp * 32 + rows] = True
    else:
      valid[:rows] = True
    # ---- forward chain, each stage from the stashed stage in front of it ----
    if not tangent:
      rows_pad = (rows + 63) // 64 * 64
      if lv == 2:
        x = bg['points'].cpu().double()
        ids = bg['warp_ids'].cpu().long()
      else:
        x = _f32rows(model, ws, 'points_raw', lv, rows_pad, 3)[:rows]
        ids = b['metadata']['warp'].reshape(-1).long().repeat_interleave(S[lv])
      want_in = _trunk_input(x, table[ids], ALPHA, F)
      _close(X[:rows, :Win], q(want_in), (name, 'trunk input'), ulp_frac=0.02)
      assert (X[:rows, Win:] == 0).all()
      masks = [(Hs[l] > 0) for l in range(6)]
      if lv == 0:
        prim_x, prim_ids, prim_masks = x, ids, masks
      wv = _f32rows(model, ws, 'w_st_wv', lv, rows_pad, 8)[:rows]
    else:
      masks = [m[:ngp * 32].repeat(3, 1) for m in prim_masks]
      for c in range(3):
        want_in = _trunk_input(prim_x, table[prim_ids], ALPHA, F, tangent_dir=c)
        _close(X[c * ngp * 32:c * ngp * 32 + rows, :Win], q(want_in), (name, 'tangent input', c), ulp_frac=0.02)
      wv = _f32rows(model, ws, 'w_st_wv', lv, 3 * ((rows + 63) // 64 * 64), 8)
      rp = (rows + 63) // 64 * 64
      wv = torch.cat([wv[c * rp:c * rp + rows] for c in range(3)], 0)
    prev = X[:, :64]
    for l in range(6):
      Wl = Wk[l]
      if l == 0:
        pre = prev[:, :Win] @ q(Wl)
      elif l == 4:
        pre = prev @ q(Wl[:128]) + X[:, :Win] @ q(Wl[128:])
      else:
        pre = prev @ q(Wl)
      w_l = q(pre * masks[l]) if tangent else q(torch.relu(pre + hilo(bk[l])))
      _close(Hs[l][valid], w_l[valid], (name, f'h{l + 1}'))
      prev = Hs[l]
    heads = Hs[5] @ q(Wh) + (0 if tangent else hilo(bh))
    hv = valid.nonzero().reshape(-1)
    got_wv = torch.cat([wv[:, 0:3], wv[:, 4:7]], 1)
    np.testing.assert_allclose(got_wv.numpy(), heads[hv].numpy(), atol=2e-5 * max(heads.abs().max().item(), 1e-3) + 1e-7, err_msg=f'{name} heads')
    # ---- reverse chain ----
    d5 = q((DH[:, :6] @ q(Wh).T) * masks[5])
    _close(DY[5][valid], d5[valid], (name, 'dpre_5'))
    for l in range(5, 0, -1):
      d = q((DY[l] @ q(Wk[l][:128]).T) * masks[l - 1])
      _close(DY[l - 1][valid], d[valid], (name, f'dpre_{l - 1}'))
    assert (DY[0][~valid] == 0).all() and (DH[~valid] == 0).all()      # padding rows carry no gradient
    # ---- the leaves' gradients: X^T dY over the rows of this pass (tangent: no bias, training.py / warping.py:385-387) ----
    for l in range(6):
      gk = (X[:, :Win] if l == 0 else Hs[l - 1]).T @ DY[l]
      if l == 4:
        gk = torch.cat([gk, X[:, :Win].T @ DY[4]], 0)
      want[f'trunk/hidden_{l}/kernel'] = want[f'trunk/hidden_{l}/kernel'] + gk
      if not tangent:
        want[f'trunk/hidden_{l}/bias'] = want[f'trunk/hidden_{l}/bias'] + DY[l].sum(0)
    want['branches_w/logit/kernel'] = want['branches_w/logit/kernel'] + Hs[5].T @ DH[:, 0:3]
    want['branches_v/logit/kernel'] = want['branches_v/logit/kernel'] + Hs[5].T @ DH[:, 3:6]
    if not tangent:
      want['branches_w/logit/bias'] = want['branches_w/logit/bias'] + DH[:, 0:3].sum(0)
      want['branches_v/logit/bias'] = want['branches_v/logit/bias'] + DH[:, 3:6].sum(0)
  for path, w in want.items():
    have = H.leaf(got, path).double().reshape(w.shape)
    scale = max(w.abs().max().item(), 1e-30)
    err = (have - w).abs().max().item() / scale
    worst = max(worst, err)
    assert err < 5e-3, (path, err, scale)
  # the GLO table's gradient: d code = the code columns of (dpre_0 . W0^T + dpre_4 . W4[128:]^T), summed per warp id over the
  # primal passes (the tangent input does not depend on the code)
  want_tab = torch.zeros_like(table)
  for name, lv, rows in (('coarse', 0, B * S[0]), ('fine', 1, B * S[1]), ('background', 2, nbg)):
    ng = (rows + 255) // 256 * 8
    DY = H.bf16_stash(model, ws, 'bw_dy', lv, 6, 4, ng * 32, ngroups=ng)
    dcode = (DY[0] @ q(Wk[0]).T + DY[4] @ q(Wk[4][128:]).T)[:rows, 3 + 6 * F:Win]
    ids = bg['warp_ids'].cpu().long() if lv == 2 else b['metadata']['warp'].reshape(-1).long().repeat_interleave(S[lv])
    want_tab.index_add_(0, ids, dcode)
  have = H.leaf(got, 'metadata_encoder/embed/embedding').double()
  err = (have - want_tab).abs().max().item() / want_tab.abs().max().item()
  assert err < 5e-3, ('embedding', err)
  print(f'[bf16 SE3 trunk given the stash, B={B}] worst leaf {worst:.2e}, embedding {err:.2e}')


@pytest.mark.parametrize('kw,head_scale', [(dict(num_warp_freqs=6, use_camera_metadata=True), 0.02),
                                           (dict(num_nerf_point_freqs=10, num_coarse_samples=64, num_fine_samples=64), 0.02),
                                           (dict(num_warp_freqs=6, use_camera_metadata=True), 1.0),
                                           # TranslationField (warping.py:62-199) = the trunk with a zero rotation head
                                           (dict(warp_field_type='translation', num_warp_freqs=5), 0.05),
                                           # codes from modules.TimeEncoder (one row per ray) instead of the GLO table
                                           (dict(warp_metadata_encoder_type='time', num_warp_freqs=6), 0.02)])
def test_bf16_warp_against_the_float32_trunk(kw, head_scale):
  """What the bfloat16 trunk costs next to the float32 trunk, everything else (bf16 NeRF MLPs, rays, uniforms) equal.

  The trunk's bf16 operands move a warped point by ~5e-4 of its displacement (rms; measured below).  Two regimes:
  * head_scale 0.02 -- displacements of ~0.1 scene units, what a capture's deformation field looks like: the perturbation is
    ~5e-5, the NeRF posenc (2^(F_p - 1)) turns it into ~1e-2 rad at the top band, and every gradient leaf keeps its direction;
  * head_scale 1 -- the oracle's "trained-like" heads, which throw points ~4 scene sizes away (built to expose indexing bugs, not
    to resemble a scene): the same relative error is 2e-3 absolute = ~1 rad at the top band, the rendered colours and with
    them the upstream gradient d loss / d x' change by O(1), and the gradient directions of the warp field decorrelate.  That is
    the 2^(F_p - 1) amplification any perturbation of x' meets (two float32 evaluation orders differ by the same factor times 1e-7,
    tests/test_gpu_pinned.py), not an error of the kernels -- those are pinned by the given-the-stash test above; here only the
    values that do not pass through the NeRF posenc are compared (warped points, regulariser values)."""
  from nerfies_amd import params as P
  time_enc = kw.get('warp_metadata_encoder_type') == 'time'
  B, nbg = 96, (0 if time_enc else 512)   # the background points carry warp ids, which a time-encoded field does not have
  wextra = {'alpha': ALPHA, 'time_alpha': 1.0} if time_enc else {'alpha': ALPHA}
  spec, p, b, model, fp, rngs, bg = _setup(B, seed=11, nbg=nbg, head_scale=head_scale, **kw)
  gb = H.gpu_batch(b)
  realistic = head_scale < 1.0
  # inference: warped points
  o16 = model.apply({'params': fp}, gb, wextra, rngs=rngs, return_points=True, bf16=True)
  o32 = model.apply({'params': fp}, gb, wextra, rngs=rngs, return_points=True, bf16='mlp')
  for lv in ('coarse', 'fine'):
    disp = (o32[lv]['warped_points'] - o32[lv]['points']).abs().max().item()
    dxs = (o16[lv]['warped_points'] - o32[lv]['warped_points']).abs()
    dx, rms = dxs.max().item(), dxs.pow(2).mean().sqrt().item()
    print(f"[bf16 vs f32 SE3 trunk, heads x {head_scale}, {lv}] max |x' - x| {disp:.4f}; |x'_bf16 - x'_f32| max {dx:.2e} rms {rms:.2e}; "
          f"max |rgb_bf16 - rgb_f32| {(o16[lv]['rgb'] - o32[lv]['rgb']).abs().max().item():.2e}")
    assert disp > 1e-3 and dx < 2e-2 * disp + 1e-5 and rms < 2e-3 * disp + 1e-6, (lv, dx, rms, disp)
    if realistic:
      assert (o16[lv]['rgb'] - o32[lv]['rgb']).abs().max().item() < 2e-2
  # training: loss, regulariser values, gradient directions
  extra = dict(warp_extra=wextra, rngs=rngs, background=bg, elastic={'weight': 0.01, 'reduce_method': 'weight'})
  g32, s32 = model.loss_and_grad(fp, gb, bf16='mlp', **extra)
  g32, s32 = g32.clone(), s32.clone()
  g16, s16 = model.loss_and_grad(fp, gb, bf16=True, **extra)
  assert torch.isfinite(g16).all() and torch.isfinite(s16).all()
  assert abs(s16[4].item() - s32[4].item()) < 1e-3 + 2e-2 * abs(s32[4].item())
  for k, what in ((5, 'background loss'), (6, 'elastic loss'), (7, 'elastic residual'))[(1 if time_enc else 0):]:
    assert abs(s16[k].item() - s32[k].item()) < 2e-2 * abs(s32[k].item()) + 1e-9, (what, s16[k].item(), s32[k].item())
  t32, t16 = P.tree_from_flat(g32.cpu(), model.layout), P.tree_from_flat(g16.cpu(), model.layout)
  cos_min, cos_warp, table = 1.0, 1.0, []
  for path, a in O.tree_leaves_with_path(t32):
    c = torch.nn.functional.cosine_similarity(a.flatten().double(), H.leaf(t16, path).flatten().double(), dim=0).item()
    table.append((path, c))
    cos_min = min(cos_min, c)
    if path.startswith('warp_field'):
      cos_warp = min(cos_warp, c)
  print(f'[bf16 vs f32 SE3 trunk {kw}, heads x {head_scale}] loss {s16[4].item():.6f} / {s32[4].item():.6f}, elastic {s16[6].item():.4e} / '
        f'{s32[6].item():.4e}, background {s16[5].item():.4e} / {s32[5].item():.4e}; min leaf cosine {cos_min:.4f} (warp field {cos_warp:.4f})')
  if realistic:
    assert cos_warp > 0.97 and cos_min > 0.97, '\n'.join(f'    {c:.4f}  {path}' for path, c in table)


def test_bf16_warp_inference_renderer_and_opt_out():
  """GraphedChunkRenderer in the bf16 mode renders through the bf16 trunk; the Jacobian output keeps the float32 trunk."""
  from nerfies_amd import evaluation
  spec, p, b, model, fp, rngs, _ = _setup(64, seed=3, use_stratified_sampling=False)
  gb = {k: v for k, v in H.gpu_batch(b).items() if k != 'rgb'}
  fn = evaluation.GraphedChunkRenderer(model, bf16=True)
  a = fn(0, 1, fp, gb, {'alpha': ALPHA})
  direct = model.apply({'params': fp}, gb, {'alpha': ALPHA}, bf16=True)
  np.testing.assert_array_equal(a['fine']['rgb'].cpu().numpy(), direct['fine']['rgb'].cpu().numpy())
  j16 = model.apply({'params': fp}, gb, {'alpha': ALPHA}, return_warp_jacobian=True, bf16=True)
  j32 = model.apply({'params': fp}, gb, {'alpha': ALPHA}, return_warp_jacobian=True)
  np.testing.assert_allclose(j16['coarse']['warp_jacobian'].cpu().numpy(), j32['coarse']['warp_jacobian'].cpu().numpy(), atol=1e-6)
