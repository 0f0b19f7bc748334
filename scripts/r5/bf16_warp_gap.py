#!/usr/bin/env python
"""How the held-out PSNR of the three precision modes develops with the step count at the vrig preset's posenc widths (F_p = 8,
F_w = 6, G = 8) on the deforming synthetic scene of tests/test_gpu_bf16_convergence.py: fp32, bf16 (NeRF MLPs + SE3 trunk) and
bf16='mlp' (SE3 trunk in float32), two sampling-key seeds each, evaluated at 600 / 2000 / 6000 steps of ONE 6000-step schedule.
  python scripts/r5/bf16_warp_gap.py > gpurun_out/r5m/bf16_warp_gap.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
from nerfies_amd import models, training
from test_gpu_bf16_convergence import _scene_rgb, _psnr
DEV = 'cuda:0'
FP, FW, B, K, NB, NID = 8, 6, 256, 6000, 64, 4


class Cfg:
  num_coarse_samples, num_fine_samples, num_nerf_point_freqs = 32, 32, FP
  sigma_activation, use_stratified_sampling, use_viewdirs = 'softplus', True, True
  use_warp, warp_field_type, num_warp_freqs, num_warp_features = True, 'se3', FW, 8


g = torch.Generator().manual_seed(0)
n_train, n_test = NB * B, 2048
o = (torch.rand(n_train + n_test, 3, generator=g) - 0.5).to(DEV)
d = torch.nn.functional.normalize(torch.randn(n_train + n_test, 3, generator=g), dim=-1).to(DEV)
ids = torch.randint(0, NID, (n_train + n_test, 1), generator=g).to(DEV)
frame_shift = (0.04 * torch.randn(NID, 3, generator=g)).to(DEV)
rgb = _scene_rgb(o, d, frame_shift[ids[:, 0]])
ecfg = type('E', (Cfg,), {'use_stratified_sampling': False})
em, _ = models.construct_nerf(7, ecfg, n_test, [0], [0], list(range(NID)), 0.05, 1.0, device=DEV)
test = {'origins': o[n_train:], 'directions': d[n_train:], 'metadata': {'warp': ids[n_train:]}}
out = {'config': f'F_p={FP} F_w={FW} B={B} 32+32 samples, lr 1e-3 -> 1e-4 over {K} steps, warp alpha 0 -> F_w over the first half, elastic 1e-3', 'runs': {}}
for mode, bf in (('f32', False), ('bf16', True), ('bf16mlp', 'mlp')):
  for key0 in (1, 1001):
    model, fp = models.construct_nerf(7, Cfg, B, [0], [0], list(range(NID)), 0.05, 1.0, device=DEV)
    state = training.TrainState(optimizer=training.Optimizer(fp))
    key, at = key0, {}
    for k in range(K):
      sp = training.ScalarParams(learning_rate=1e-3 * 0.1 ** (k / K), elastic_loss_weight=1e-3)
      state = state.replace(warp_alpha=float(FW) * min(1.0, k / (0.5 * K)))
      i0 = (k % NB) * B
      batch = {'origins': o[i0:i0 + B], 'directions': d[i0:i0 + B], 'rgb': rgb[i0:i0 + B], 'metadata': {'warp': ids[i0:i0 + B]}}
      state, stats, key = training.train_step(model, key, state, batch, sp, use_elastic_loss=True, elastic_reduce_method='weight', bf16=bf)
      if k + 1 in (600, 2000, 6000):
        at[k + 1] = _psnr(em.apply({'params': fp}, test, {'alpha': state.warp_alpha})['fine']['rgb'], rgb[n_train:])
    out['runs'][f'{mode}/{key0}'] = at
    print(mode, key0, at, file=sys.stderr, flush=True)
print(json.dumps(out))
