#!/usr/bin/env python
"""bf16 training path vs the float64 oracle with the same roundings: per-leaf error table."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from oracle import nerfies_oracle as O
import helpers as H
import test_gpu_bf16_train as T
from nerfies_amd import params as P

B = int(sys.argv[1]) if len(sys.argv) > 1 else 37
spec, p, b, t_rand, u, model, fp, rngs = T._setup(B)
grad, stats = model.loss_and_grad(fp, H.gpu_batch(b), rngs=rngs, bf16=True)
torch.cuda.synchronize()
ws = model.workspace(B, True, H.DEV, bf16=True)
S1 = spec.num_coarse_samples + spec.num_fine_samples
z_fine = torch.from_numpy(H._ws_words(model, ws, 'z', 1, B * S1).view('float32').reshape(B, S1).copy()).double()
with H.host_threads(64), O.dense_hook(T.bf16_dense_for(spec)):
  loss, ostats, ograds, ret = O.loss_and_grad(p, spec, b, t_rand=t_rand, u=u, fixed_fine_z=z_fine)
print('loss', stats[4].item(), loss.item())
got = P.tree_from_flat(grad.cpu(), model.layout)
for path, og in O.tree_leaves_with_path(ograds):
  a = H.leaf(got, path).double()
  sc = max(og.abs().max().item(), 1e-30)
  d = (a - og).abs()
  print(f'{path:50s} max {sc:9.3e} err/max {d.max().item() / sc:8.2e} relL2 {(a - og).norm().item() / og.norm().item():8.2e}')

# ---- forward only: bf16 inference forward vs the rounded oracle, per-sample weights / rendered outputs ----
out = model.apply({'params': fp}, H.gpu_batch(b), {}, rngs=rngs, return_weights=True, bf16=True, return_z_vals=True)
out32 = model.apply({'params': fp}, H.gpu_batch(b), {}, rngs=rngs, return_weights=True, return_z_vals=True)
for lv in ('coarse', 'fine'):
  for k in ('rgb', 'acc', 'depth', 'weights'):
    a = out[lv][k].double().cpu(); o = ret[lv][k].detach(); f = out32[lv][k].double().cpu()
    print(f'{lv}/{k:8s} |bf16 gpu - rounded oracle| {(a - o).abs().max().item():.2e}   |bf16 gpu - fp32 gpu| {(a - f).abs().max().item():.2e}   |z| {(out[lv]["z_vals"].double().cpu() - ret[lv]["z_vals"]).abs().max().item():.1e}')
