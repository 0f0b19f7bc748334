"""The float32 NeRF chain on 32-row tiles (csrc/mlp_chain32.hip, NRF_OPT_CHAIN_TILE_ROWS = 32) against the 64-row kernels
(csrc/mlp_chain.hip) on the same inputs (modules.py:26-62, 65-169; models.py:270-277).

Both tilings write the SAME HBM images (64-row fragment-order stash, sign-bit words, d raw / out4 rows), and every output
element is the same fmaf chain in both (bias, then k = 0, 1, ... through the same MFMA k-steps; the alpha head and the logits
as the same four partial sums), so stashes, ReLU bits, rendered values and the stashed gradients must agree BIT FOR BIT --
which is what lets the library pick the tiling per launch without a ray's result depending on the size of the launch it
rides in (tests/test_gpu_fullsize.py).  Only float atomics (bias / per-ray condition sums) differ in order.  Shapes are chosen so
that the last 64-row tile is ragged in both ways: more than 32 valid rows (second half partly padding) and fewer (second
half all padding)."""
import ctypes as C

import numpy as np
import pytest
import torch

import helpers as H
from oracle import nerfies_oracle as O

pytestmark = pytest.mark.gpu


def _region(model, ws, name, level, nfloats):
  from nerfies_amd import lib as L
  off = C.c_int64(0)
  L.check(model.lib.nrf_debug_ws_offset(model.handle, name.encode(), level, C.byref(off)), model.lib)
  return ws[off.value:off.value + nfloats].clone()


def _train_once(rows_opt, spec, oparams, batch, B, regions, **kw):
  model, fp = H.gpu_model(spec, oparams, B)
  model.set_chain_tile_rows(rows_opt)
  grad, stats = model.loss_and_grad(fp, batch, **kw)
  torch.cuda.synchronize()
  ws = model.workspace(B, True, H.DEV)
  out = {'grad': grad.clone(), 'stats': stats.clone()}
  for lv, S in ((0, spec.num_coarse_samples), (1, spec.num_coarse_samples + spec.num_fine_samples)):
    nt = (B * S + 63) // 64
    for name, per_tile in regions.items():
      out[(name, lv)] = _region(model, ws, name, lv, nt * per_tile)
  return out, model


@pytest.mark.parametrize('B,nc,nf', [(37, 24, 40), (33, 24, 8), (5, 64, 128)])
def test_stash_bits_and_gradients_match_the_64_row_kernels(B, nc, nf):
  spec = O.ModelSpec(num_coarse_samples=nc, num_fine_samples=nf, num_nerf_point_freqs=8, use_stratified_sampling=False,
                     use_camera_metadata=True)
  oparams = O.init_params(spec, seed=11, trained_like=True, dtype=torch.float32)
  batch = H.gpu_batch(O.synthetic_batch(B, seed=12, dtype=torch.float32))
  regions = {'st_pe': 64 * 64, 'st_h': 8 * 256 * 64, 'st_bn': 256 * 64, 'st_rgbh': 128 * 64, 'bits_trunk': 8 * 4 * 128, 'bits_rgbh': 4 * 64,
             'out4': 64 * 4, 'dy_trunk': 8 * 256 * 64, 'dy_bn': 256 * 64, 'dy_rgbh': 128 * 64}
  a, _ = _train_once(64, spec, oparams, batch, B, regions)
  b, _ = _train_once(32, spec, oparams, batch, B, regions)
  # every stashed activation, every sign bit, the kernel outputs and every stashed pre-activation gradient: the same fmaf chains in
  # both tilings (trunk / bottleneck / rgb hidden: bias, then k = 0, 1, ... through the same MFMA k-steps; alpha head and logits: the
  # same four partial sums combined in the same order) -> the same bits, at both levels (the fine depths derive from the coarse
  # weights, which are therefore the same too)
  for lv in (0, 1):
    # st_h / dy_trunk are [layer][tile][...]: the debug offset is the level's base and the level's layers are contiguous behind it
    for name in regions:
      assert torch.equal(a[(name, lv)].view(torch.int32), b[(name, lv)].view(torch.int32)), (name, lv)
  assert torch.equal(a['stats'][:5], b['stats'][:5])
  # gradients: the reverse tilings differ in the ORDER of their float atomics only (bias column sums, per-ray condition sums)
  ga, gb = a['grad'], b['grad']
  model, _ = H.gpu_model(spec, oparams, B)
  for name, off, shape in model.layout.entries:
    n = int(np.prod(shape))
    x, y = ga[off:off + n], gb[off:off + n]
    assert (x - y).abs().max().item() <= 1e-3 * x.abs().max().item() + 1e-12, name


def test_mixed_tilings_forward_32_reverse_64_with_the_warp_field():
  """Models with a warp field run the 32-row FORWARD and keep the 64-row reverse chain (the 32-row one has no d-points path):
  the two must meet in the stash.  Gradients incl. the warp leaves against the all-64 run."""
  spec = O.ModelSpec(num_coarse_samples=24, num_fine_samples=40, num_nerf_point_freqs=6, use_stratified_sampling=False, use_warp=True,
                     num_warp_freqs=4)
  oparams = O.init_params(spec, seed=21, trained_like=True, dtype=torch.float32)
  B = 29
  batch = H.gpu_batch(O.synthetic_batch(B, seed=22, dtype=torch.float32))
  a, model = _train_once(64, spec, oparams, batch, B, {}, warp_extra={'alpha': 2.5, 'time_alpha': 0.0})
  b, _ = _train_once(32, spec, oparams, batch, B, {}, warp_extra={'alpha': 2.5, 'time_alpha': 0.0})
  assert torch.allclose(a['stats'], b['stats'], rtol=1e-5, atol=1e-7)
  for name, off, shape in model.layout.entries:
    n = int(np.prod(shape))
    x, y = a['grad'][off:off + n], b['grad'][off:off + n]
    assert (x - y).abs().max().item() <= 5e-4 * x.abs().max().item() + 1e-12, name


@pytest.mark.parametrize('rows_opt', [32, 64])
def test_inference_outputs_against_the_oracle(rows_opt):
  """NerfModel.apply (models.py:289-375) under either tiling against the float64 oracle: rendered rgb / depth / acc to 1e-4."""
  spec = O.ModelSpec(num_coarse_samples=64, num_fine_samples=128, num_nerf_point_freqs=8, use_stratified_sampling=False)
  oparams = O.init_params(spec, seed=3, trained_like=True, dtype=torch.float32)
  B = 21
  ob = O.synthetic_batch(B, seed=4, dtype=torch.float32)
  model, fp = H.gpu_model(spec, oparams, B)
  model.set_chain_tile_rows(rows_opt)
  out = model.apply({'params': fp}, {k: v for k, v in H.gpu_batch(ob).items() if k != 'rgb'}, {'alpha': 0.0})
  p64 = O.tree_map(lambda t: t.double(), oparams)
  b64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in ob.items()}
  ref = O.nerf_model_apply(p64, spec, b64)
  for lv in ('coarse', 'fine'):
    for k in ('rgb', 'depth', 'acc'):
      err = (out[lv][k].double().cpu() - ref[lv][k]).abs().max().item()
      assert err <= 1e-4, (rows_opt, lv, k, err)


def test_option_is_validated_and_replans():
  from nerfies_amd import lib as L
  spec = O.ModelSpec(num_coarse_samples=8, num_fine_samples=8, num_nerf_point_freqs=4, use_stratified_sampling=False)
  oparams = O.init_params(spec, seed=1, trained_like=True, dtype=torch.float32)
  model, fp = H.gpu_model(spec, oparams, 4)
  with pytest.raises(L.NrfError):
    model.set_chain_tile_rows(48)
  with pytest.raises(L.NrfError):
    L.check(model.lib.nrf_set_option(model.handle, 99, 0), model.lib)
  batch = H.gpu_batch(O.synthetic_batch(4, seed=2, dtype=torch.float32))
  grads = []
  for rows in (64, 32, 0, 32, 64):   # one handle, one workspace, the option changed between steps
    model.set_chain_tile_rows(rows)
    g, _ = model.loss_and_grad(fp, batch)
    grads.append(g.clone())
  for g in grads[1:]:
    assert (g - grads[0]).abs().max().item() <= 2e-4 * grads[0].abs().max().item()
