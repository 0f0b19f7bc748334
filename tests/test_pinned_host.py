"""CPU checks of the pinned-branch test machinery (tests/helpers.py): the sign-bit decoder against a Python restatement of
the kernels' epilogue layout (csrc/chain_common.h fwd_epilogue), and the oracle's relu hook."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import helpers as H  # noqa: E402
from oracle import nerfies_oracle as O  # noqa: E402


def _encode(mask, ncb):
  """mask bool [L][rows_pad][4*32*ncb] -> uint32 words in the kernels' order: bits_wave[lane * NCB + cb], nibble q,
  bit e <-> tile row 4 * q_granule(q, h) + e, feature wave*32*NCB + 32*cb + (lane & 31)."""
  L, rows, W = mask.shape
  nt = rows // 64
  words = np.zeros((L, nt, 4, 64, ncb), np.uint32)
  for l in range(L):
    for t in range(nt):
      for wave in range(4):
        for lane in range(64):
          j, h = lane & 31, lane >> 5
          for cb in range(ncb):
            n = wave * 32 * ncb + 32 * cb + j
            w = 0
            for q in range(8):
              g = (q & 1) + 2 * h + 4 * (q >> 1)
              for e in range(4):
                if mask[l, t * 64 + 4 * g + e, n]:
                  w |= 1 << (4 * q + e)
            words[l, t, wave, lane, cb] = w
  return words


@pytest.mark.parametrize('ncb', [1, 2])
def test_sign_bit_decoder_inverts_the_kernel_layout(ncb):
  rng = np.random.default_rng(ncb)
  mask = rng.random((2, 128, 128 * ncb)) < 0.5
  got = H._decode_bits(_encode(mask, ncb).reshape(-1), 2, 2, ncb, 100)
  assert got.shape == (2, 100, 128 * ncb)
  np.testing.assert_array_equal(got.numpy(), mask[:, :100])


def test_relu_hook_with_own_signs_is_the_identity():
  """Pinning the oracle to ITS OWN branch pattern must reproduce its gradients bit for bit; pinning a flipped unit must
  change only what depends on it."""
  spec = O.ModelSpec(num_coarse_samples=8, num_fine_samples=8, num_nerf_point_freqs=2, use_warp=True, num_warp_freqs=2)
  p = O.init_params(spec, seed=1, trained_like=True)
  b = O.synthetic_batch(3, seed=2)
  loss0, _, g0, _ = O.loss_and_grad(p, spec, b, warp_alpha=2.0, use_elastic_loss=True, elastic_loss_weight=0.01)
  seen = {}

  def own(name, layer, pre):
    seen.setdefault(name, set()).add(layer)
    return pre * (pre.detach() > 0).to(pre.dtype)
  with O.relu_hook(own):
    loss1, _, g1, _ = O.loss_and_grad(p, spec, b, warp_alpha=2.0, use_elastic_loss=True, elastic_loss_weight=0.01)
  assert O._RELU_HOOK is None
  assert sorted(seen) == ['coarse/MLP_0', 'coarse/MLP_1', 'coarse/warp', 'fine/MLP_0', 'fine/MLP_1', 'fine/warp']
  assert seen['coarse/MLP_0'] == set(range(8)) and seen['coarse/warp'] == set(range(6)) and seen['fine/MLP_1'] == {0}
  assert abs(loss0.item() - loss1.item()) < 1e-14
  for (path, a), (_, c) in zip(O.tree_leaves_with_path(g0), O.tree_leaves_with_path(g1)):
    np.testing.assert_allclose(c.numpy(), a.numpy(), rtol=1e-10, atol=1e-14, err_msg=path)


def test_pinned_relu_counts_disagreements():
  pre = torch.tensor([[1.0, -1.0, 1e-9, -1e-9]], dtype=torch.float64)
  hook = H.PinnedRelu({'x': [torch.tensor([[True, False, False, True]])]})
  out = hook('x', 0, pre)
  np.testing.assert_array_equal(out.numpy(), [[1.0, 0.0, 0.0, -1e-9]])
  assert hook.flips == 2 and hook.total == 4 and hook.worst < 1e-8 and hook.quantile(0.5) < 1e-8


def _fp32_vs_fp64_pinned(Fp, B=9, alpha=8.0, seed=3):
  """Worst per-leaf gradient error (relative to the leaf's max-abs) of the oracle's float32 evaluation against its
  float64 one, with the float64 run pinned to the float32 run's ReLU branches and fine samples -- the same comparison
  tests/test_gpu_pinned.py makes for the HIP path."""
  spec = O.ModelSpec(use_warp=True, use_stratified_sampling=True, num_nerf_point_freqs=Fp, num_coarse_samples=32, num_fine_samples=32)
  p64 = O.init_params(spec, seed=seed, trained_like=True, dtype=torch.float64)
  b64 = O.synthetic_batch(B, seed=seed + 1, dtype=torch.float64)
  g = torch.Generator().manual_seed(seed)
  t_rand, u = torch.rand(B, 32, generator=g), torch.rand(B, 32, generator=g)
  f32 = lambda t: t.float() if torch.is_tensor(t) and t.is_floating_point() else t
  p32 = O.tree_map(f32, p64)
  b32 = {k: (O.tree_map(f32, v) if isinstance(v, dict) else f32(v)) for k, v in b64.items()}
  masks = {}

  def record(name, layer, pre):
    m = pre.detach() > 0
    masks.setdefault(name, {}).setdefault(layer, m.reshape(-1, m.shape[-1]))
    return pre * m.to(pre.dtype)
  with O.relu_hook(record):
    _, _, g32, r32 = O.loss_and_grad(p32, spec, b32, warp_alpha=alpha, t_rand=t_rand, u=u)
  hook = H.PinnedRelu({k: [v[i] for i in sorted(v)] for k, v in masks.items()})
  with O.relu_hook(hook):
    _, _, g64, _ = O.loss_and_grad(p64, spec, b64, warp_alpha=alpha, t_rand=t_rand.double(), u=u.double(),
                                   fixed_fine_z=r32['fine']['z_vals'].double())
  assert hook.flips <= 1e-3 * hook.total
  return max((a.double() - b).abs().max().item() / b.abs().max().item()
             for (_, a), (_, b) in zip(O.tree_leaves_with_path(g32), O.tree_leaves_with_path(g64)))


def test_float32_floor_of_the_warp_gradients():
  """Why tests/helpers.py grad_tol() is 4e-3 at F_p = 10: ANY float32 evaluation of the warp-on gradient is ~1-2e-3 from
  float64 there (and ~4e-4 at F_p = 8, ~1e-5 at F_p = 4) even with identical ReLU branches and identical fine samples:
  the error grows like 2^F_p (phase error of the top posenc band from the float32 rounding of the warped point)."""
  e4, e8, e10 = _fp32_vs_fp64_pinned(4), _fp32_vs_fp64_pinned(8), _fp32_vs_fp64_pinned(10)
  print(f'float32 restatement vs float64, pinned: F_p=4 {e4:.1e}  F_p=8 {e8:.1e}  F_p=10 {e10:.1e}')
  assert e4 < 1e-4 and e8 < 2e-3
  assert 5e-4 < e10 < 4e-3      # a 2e-3 bound would sit inside float32's own noise at F_p = 10
  assert e10 > 2 * e8 > 4 * e4
