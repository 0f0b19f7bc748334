#!/usr/bin/env python
"""bf16 training path vs the fp32 path on the same rays: loss and per-leaf gradient error (relative to the leaf's max-abs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from oracle import nerfies_oracle as O
import helpers as H
from nerfies_amd import params as P

B = int(sys.argv[1]) if len(sys.argv) > 1 else 40
spec = O.ModelSpec(num_coarse_samples=64, num_fine_samples=128, num_nerf_point_freqs=8, use_stratified_sampling=True)
p = O.init_params(spec, seed=3, trained_like=True)
b = O.synthetic_batch(B, seed=4)
model, fp = H.gpu_model(spec, p, B)
gb = H.gpu_batch(b)
g = torch.Generator().manual_seed(5)
rngs = {'coarse': torch.rand(B, 64, generator=g).to(H.DEV), 'fine': torch.rand(B, 128, generator=g).to(H.DEV)}
g32, s32 = model.loss_and_grad(fp, gb, rngs=rngs)
g32, s32 = g32.clone(), s32.clone()
g16, s16 = model.loss_and_grad(fp, gb, rngs=rngs, bf16=True)
torch.cuda.synchronize()
print('stats fp32', s32[:5].tolist())
print('stats bf16', s16[:5].tolist())
t32, t16 = P.tree_from_flat(g32.cpu(), model.layout), P.tree_from_flat(g16.cpu(), model.layout)
for path, a in O.tree_leaves_with_path(t32):
  bq = H.leaf(t16, path)
  sc = max(a.abs().max().item(), 1e-30)
  print(f'{path:55s} max {sc:9.3e}  err/max {(bq - a).abs().max().item() / sc:8.2e}  cos {torch.nn.functional.cosine_similarity(a.flatten(), bq.flatten(), dim=0).item():.5f}  nan {int(torch.isnan(bq).sum())}')
