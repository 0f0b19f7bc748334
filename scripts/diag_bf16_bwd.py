#!/usr/bin/env python
"""Localises a bf16 backward mismatch: recomputes the dgrad chain in float64 from the kernels' own stash and compares each
stashed dpre and each weight gradient block-wise."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from oracle import nerfies_oracle as O
import helpers as H
import test_gpu_bf16_train as T
from nerfies_amd import params as P

B, kw = T.CASES[int(sys.argv[1]) if len(sys.argv) > 1 else 2]
spec, p, b, t_rand, u, model, fp, rngs = T._setup(B, **kw)
grad, stats = model.loss_and_grad(fp, H.gpu_batch(b), rngs=rngs, bf16=True)
torch.cuda.synchronize()
ws = model.workspace(B, True, H.DEV, bf16=True)
S = (spec.num_coarse_samples, spec.num_coarse_samples + spec.num_fine_samples)
got = P.tree_from_flat(grad.cpu(), model.layout)
q = H.bf16_round
tw = spec.nerf_trunk_width
for lv, name in enumerate(('nerf_mlps_coarse', 'nerf_mlps_fine')):
  rows = B * S[lv]
  prm = O.tree_map(lambda t: t.float().double(), p[name])
  W = lambda path: H.leaf(prm, path)
  h = [t[:, :tw] for t in H.bf16_stash(model, ws, 'b_h', lv, 8, 8, rows)]
  dy = [t[:, :tw] for t in H.bf16_stash(model, ws, 'b_dy', lv, 8, 8, rows)]
  dbn = H.bf16_stash(model, ws, 'b_dbn', lv, 1, 8, rows)[0][:, :tw]
  draw = H.bf16_stash(model, ws, 'b_dsmall', lv, 1, 2, rows)[0][:, :4]
  dsig = draw[:, 3:4]
  wa = W('MLP_2/logit/kernel')[:tw]; wa = q(wa) + q(wa - q(wa))
  d = q((dbn @ q(W('bottleneck/kernel')).T + dsig @ wa.T) * (h[7] > 0))
  for l in range(7, -1, -1):
    e = (dy[l] - d).abs()
    bad = (e > 1e-2 * d.abs().max()).nonzero()
    print(f'{name} dpre_{l}: stash vs float64 chain max err {e.max().item() / d.abs().max().item():.2e}  bad elements {len(bad)}'
          + (f' rows {sorted(set((bad[:, 0] // 32).tolist()))[:12]} (groups of 32) cols {sorted(set((bad[:, 1] // 32).tolist()))}' if len(bad) else ''))
    x = h[l - 1] if l > 0 else None
    if l > 0:
      have = H.leaf(got[name], f'MLP_0/hidden_{l}/kernel').double()[:tw]
      want = x.T @ dy[l]          # from the STASHED dY: isolates the wgrad kernel
      e2 = (have - want).abs() / want.abs().max()
      blk = e2.reshape(tw // 32, 32, tw // 32, 32).amax((1, 3))
      print(f'   wgrad hidden_{l}: max err {e2.max().item():.2e}; 32x32 blocks above 1e-3: {(blk > 1e-3).nonzero().tolist()[:16]}')
      d = q((dy[l] @ q(W(f"MLP_0/hidden_{l}/kernel")[:tw]).T) * (h[l - 1] > 0))
