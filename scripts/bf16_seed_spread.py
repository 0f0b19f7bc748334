#!/usr/bin/env python
"""How far apart do training runs of the config-A shape land that differ ONLY in rounding / sampling keys?  N seeds x {fp32, bf16}
(2000 Adam steps each, the protocol of scripts/bf16_convergence.py), held-out PSNR rendered by the fp32 eval path.  Run once per
library build (NRF_LIB_PATH selects a variant) to compare two builds of the bf16 kernels on the same seeds.
    SEEDS=5 python scripts/bf16_seed_spread.py [tag]  ->  gpurun_out/bf16_seed_spread_<tag>.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from nerfies_amd import models, training

DEV = 'cuda:0'
B, K, NS = 1024, int(os.environ.get('STEPS', 2000)), int(os.environ.get('SEEDS', 4))
MODES = os.environ.get('MODES', 'f32 bf16').split()
tag = sys.argv[1] if len(sys.argv) > 1 else 'run'


class Cfg:
  num_coarse_samples, num_fine_samples, num_nerf_point_freqs = 64, 128, 8
  sigma_activation, use_stratified_sampling, use_viewdirs = 'softplus', True, True


def scene_rgb(o, d):
  return torch.sigmoid(torch.stack([2.0 * torch.sin(3.0 * o[:, 0] + 2.0 * d[:, 1]), 2.0 * torch.cos(2.0 * o[:, 1] - 3.0 * d[:, 2]),
                                    1.5 * torch.sin(4.0 * o[:, 2] + d[:, 0])], -1))


g = torch.Generator().manual_seed(0)
nb = 128
o = (torch.rand(nb * B + 8192, 3, generator=g) - 0.5).to(DEV)
d = torch.nn.functional.normalize(torch.randn(nb * B + 8192, 3, generator=g), dim=-1).to(DEV)
rgb = scene_rgb(o, d)
em, _ = models.construct_nerf(7, type('E', (Cfg,), {'use_stratified_sampling': False}), 8192, [0], [0], [0], 0.05, 1.0, device=DEV)
test = {'origins': o[nb * B:], 'directions': d[nb * B:], 'metadata': {}}
res = {m: [] for m in MODES}
for s in range(NS):
  for mode in MODES:
    model, fp = models.construct_nerf(7, Cfg, B, [0], [0], [0], 0.05, 1.0, device=DEV)
    state = training.TrainState(optimizer=training.Optimizer(fp))
    key = 1 + 1000 * s
    for k in range(K):
      sp = training.ScalarParams(learning_rate=1e-3 * (0.1 ** (k / K)))
      i0 = (k % nb) * B
      batch = {'origins': o[i0:i0 + B], 'directions': d[i0:i0 + B], 'rgb': rgb[i0:i0 + B], 'metadata': {}}
      state, stats, key = training.train_step(model, key, state, batch, sp, bf16=(mode == 'bf16'))
    r = em.apply({'params': fp}, test, {})
    res[mode].append(float(-10.0 * np.log10(((r['fine']['rgb'] - rgb[nb * B:]) ** 2).mean().item())))
    print(f'seed {s} {mode:5s} PSNR {res[mode][-1]:.3f} dB', flush=True)
out = {'tag': tag, 'lib': os.environ.get('NRF_LIB_PATH', 'product build'), 'steps': K, 'psnr': res,
       'mean': {m: float(np.mean(v)) for m, v in res.items()}, 'std': {m: float(np.std(v)) for m, v in res.items()}}
print(json.dumps({k: out[k] for k in ('tag', 'mean', 'std')}))
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', f'bf16_seed_spread_{tag}.json'), 'w'), indent=1)
