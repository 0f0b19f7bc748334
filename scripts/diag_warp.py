"""Diagnostic: per-leaf gradient error of the warp path vs the fp64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
import helpers as H
from oracle import nerfies_oracle as O
from nerfies_amd import params as P
B = int(sys.argv[1]) if len(sys.argv) > 1 else 9
NF = int(sys.argv[2]) if len(sys.argv) > 2 else 32
skw = dict(num_coarse_samples=32, num_fine_samples=NF, num_nerf_point_freqs=8, use_stratified_sampling=False,
           use_warp=True, num_warp_freqs=8, num_warp_features=8, num_warp_embeddings=4)
spec = O.ModelSpec(**skw)
op = O.init_params(spec, seed=3, trained_like=True, dtype=torch.float64)
batch = O.synthetic_batch(B, seed=4, dtype=torch.float64)
model, fp = H.gpu_model(spec, op, B)
gb = H.gpu_batch(batch)
for rep in range(2):
  grad, stats = model.loss_and_grad(fp, gb, warp_extra={'alpha': 3.5})
  torch.cuda.synchronize()
  loss, ostats, og, _ = O.loss_and_grad(op, spec, batch, warp_alpha=3.5)
  got = P.tree_from_flat(grad.cpu(), model.layout)
  print('rep', rep, 'loss', stats[4].item(), loss.item())
  for path, g in O.tree_leaves_with_path(og):
    node = got
    for k in path.split('/'): node = node[k]
    scale = max(g.abs().max().item(), 1e-12)
    err = (node.double() - g).abs().max().item() / scale
    flag = ' <<<' if err > 1e-3 else ''
    print(f'  {path:55s} scale {scale:.3e} relerr {err:.2e}{flag}')
import numpy as np
for name in ('hidden_1', 'hidden_0', 'hidden_4'):
  g = dict(O.tree_leaves_with_path(og))[f'warp_field/trunk/{name}/kernel']
  a = got['warp_field']['trunk'][name]['kernel'].double()
  e = (a - g).abs()
  R, Cn = e.shape
  print(name, 'shape', tuple(e.shape))
  for r0 in range(0, R, 32):
    print('  rows %3d..: ' % r0 + ' '.join('%.1e' % e[r0:r0+32, c0:c0+32].max().item() for c0 in range(0, Cn, 32)),
          '| ratio', ' '.join('%.2f' % (a[r0:r0+32, c0:c0+32].norm() / max(g[r0:r0+32, c0:c0+32].norm(), 1e-30)).item() for c0 in range(0, Cn, 32)))
