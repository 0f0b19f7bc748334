"""NRF_FLAG_BF16 inference mode (mlp_bf16.hip): bfloat16 MLP operands, fp32 accumulation and compositing.  There is no
reference counterpart (BASELINE config D names it as a new-framework option), so the checks are (a) against the fp64
oracle and the fp32 HIP path at bf16-sized tolerances, (b) against the oracle run on bf16-ROUNDED weights, which isolates
the kernel's own arithmetic from the weight quantisation, (c) structural: ragged sizes, the training flag (the bf16
training path proper: tests/test_gpu_bf16_train.py)."""
import numpy as np
import pytest
import torch

import helpers as H
from oracle import nerfies_oracle as O

pytestmark = pytest.mark.gpu


def _setup(B, seed=0, **kw):
  skw = dict(num_coarse_samples=32, num_fine_samples=64, num_nerf_point_freqs=8, use_stratified_sampling=False)
  skw.update(kw)
  spec = O.ModelSpec(**skw)
  oparams = O.init_params(spec, seed=seed, trained_like=True, dtype=torch.float64)
  batch = O.synthetic_batch(B, seed=seed + 1, dtype=torch.float64)
  model, fp = H.gpu_model(spec, oparams, B)
  return spec, model, fp, H.gpu_batch(batch), oparams, batch


@pytest.mark.parametrize('B,kw', [(64, {}), (37, dict(use_camera_metadata=True)), (200, dict(num_nerf_point_freqs=10, sigma_activation='relu'))])
def test_bf16_forward_close_to_fp32_and_oracle(B, kw):
  spec, model, fp, gb, p64, b64 = _setup(B, **kw)
  lo = model.apply({'params': fp}, gb, {}, bf16=True, return_weights=True)
  hi = model.apply({'params': fp}, gb, {}, return_weights=True)
  ref = O.nerf_model_apply(p64, spec, b64)
  for lv in ('coarse', 'fine'):
    for k, tol in (('rgb', 2e-2), ('acc', 2e-2), ('depth', 2e-2)):
      a = lo[lv][k].cpu().double()
      assert torch.isfinite(a).all()
      assert (a - ref[lv][k]).abs().max().item() < tol, (lv, k, (a - ref[lv][k]).abs().max().item())
      assert (a - hi[lv][k].cpu().double()).abs().max().item() < tol
  # and it IS a different arithmetic: not bit-identical with the fp32 path
  assert not torch.equal(lo['fine']['rgb'], hi['fine']['rgb'])


def test_bf16_kernel_arithmetic_is_exact_up_to_activation_rounding():
  """Oracle evaluated with the weights rounded to bfloat16 (biases keep 16 bits via the hi+lo pair): what remains is the
  rounding of the activations to bf16 between layers, ~2^-9 relative per layer."""
  spec, model, fp, gb, p64, b64 = _setup(96, seed=3)
  rounded = O.tree_map(lambda t: t, p64)
  for path, t in O.tree_leaves_with_path(rounded):
    if path.endswith('/kernel') and path.startswith('nerf_mlps'):
      t.copy_(t.float().to(torch.bfloat16).double())
  lo = model.apply({'params': fp}, gb, {}, bf16=True)
  ref_q = O.nerf_model_apply(rounded, spec, b64)
  ref = O.nerf_model_apply(p64, spec, b64)
  err_q = (lo['fine']['rgb'].cpu().double() - ref_q['fine']['rgb']).abs().max().item()
  err = (lo['fine']['rgb'].cpu().double() - ref['fine']['rgb']).abs().max().item()
  assert err_q < 1e-2 and err_q <= err + 2e-3, (err_q, err)


def test_bf16_composes_with_training_and_the_renderer():
  from nerfies_amd import evaluation, lib as L, training
  spec, model, fp, gb, _, _ = _setup(70)
  # the training flag keeps the bf16 stash (tests/test_gpu_bf16_train.py): same forward arithmetic, and nrf_backward follows
  tr = model.apply({'params': fp}, gb, {}, train=True, bf16=True)
  inf = model.apply({'params': fp}, gb, {}, bf16=True)
  for lv in ('coarse', 'fine'):
    np.testing.assert_allclose(tr[lv]['rgb'].cpu().numpy(), inf[lv]['rgb'].cpu().numpy(), atol=1e-6)
  tr = model.apply({'params': fp}, gb, {}, train=True, bf16=True)
  d = torch.full_like(tr['fine']['rgb'], 0.01)
  g16 = model.backward({'params': fp}, gb, d, d).clone()
  model.apply({'params': fp}, gb, {}, train=True)
  g32 = model.backward({'params': fp}, gb, d, d)
  assert torch.isfinite(g16).all()
  assert torch.nn.functional.cosine_similarity(g16, g32, dim=0).item() > 0.99
  rays = {'origins': gb['origins'].reshape(7, 10, 3), 'directions': gb['directions'].reshape(7, 10, 3)}
  state = training.TrainState(optimizer=training.Optimizer(fp))
  img = evaluation.render_image(state, rays, evaluation.GraphedChunkRenderer(model, bf16=True), 1, 0, chunk=32)
  whole = model.apply({'params': fp}, gb, {}, bf16=True)
  np.testing.assert_allclose(img['rgb'].reshape(-1, 3).cpu().numpy(), whole['fine']['rgb'].cpu().numpy(), atol=1e-6)
