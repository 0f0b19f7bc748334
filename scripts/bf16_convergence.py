#!/usr/bin/env python
"""SURVEY 8d gate of the bf16 mode at the BASELINE config-A shape: 2000 Adam steps on a synthetic scene (1024 rays x (64+128),
F_p = 8, stratified), the same ray stream and init for fp32 (two runs that differ only in their sampling keys: the spread two
float32 runs have among themselves) and bf16; loss curve every 20 steps + held-out PSNR rendered by the fp32 and bf16 eval
paths.  Writes gpurun_out/bf16_convergence.json (copied to profiles/r02_bf16_convergence.json)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from nerfies_amd import models, training

DEV = 'cuda:0'
B, K = 1024, int(os.environ.get('STEPS', 2000))


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
out = {'config': 'BASELINE config A shape: 1024 rays x (64+128), F_p=8, stratified; Adam lr 1e-3 -> 1e-4 exponential; synthetic view-dependent scene', 'steps': K, 'runs': {}}
for mode, key0 in (('f32', 1), ('f32_other_keys', 1001), ('bf16', 1)):
  model, fp = models.construct_nerf(7, Cfg, B, [0], [0], [0], 0.05, 1.0, device=DEV)
  state = training.TrainState(optimizer=training.Optimizer(fp))
  key, curve = key0, []
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for k in range(K):
    sp = training.ScalarParams(learning_rate=1e-3 * (0.1 ** (k / K)))
    i0 = (k % nb) * B
    batch = {'origins': o[i0:i0 + B], 'directions': d[i0:i0 + B], 'rgb': rgb[i0:i0 + B], 'metadata': {}}
    state, stats, key = training.train_step(model, key, state, batch, sp, bf16=(mode == 'bf16'))
    if (k + 1) % 20 == 0:
      curve.append(stats['fine']['loss/rgb'])
  torch.cuda.synchronize(); dt = time.perf_counter() - t0
  psnr = {}
  for tag, kw in (('f32', {}), ('bf16', dict(bf16=True))):
    r = em.apply({'params': fp}, test, {}, **kw)
    psnr[tag] = float(-10.0 * np.log10(((r['fine']['rgb'] - rgb[nb * B:]) ** 2).mean().item()))
  curve = torch.stack(curve).cpu().numpy()
  out['runs'][mode] = {'train_seconds': dt, 'rays_per_s_incl_host': K * B / dt, 'held_out_psnr_rendered_f32': psnr['f32'],
                       'held_out_psnr_rendered_bf16': psnr['bf16'], 'loss_fine_every_20_steps': [float(x) for x in curve]}
  print(f'{mode:16s} {dt:6.1f} s  PSNR {psnr["f32"]:.3f} dB (bf16-rendered {psnr["bf16"]:.3f})  loss[100] {curve[4]:.5f} loss[1000] {curve[len(curve)//2 - 1]:.5f} loss[end] {curve[-1]:.6f}')
a, b, c = (np.array(out['runs'][m]['loss_fine_every_20_steps']) for m in ('f32', 'f32_other_keys', 'bf16'))
tail = slice(len(a) // 2, None)
out['summary'] = {'max_rel_curve_gap_f32_vs_f32_other_keys_2nd_half': float(np.abs(a[tail] - b[tail]).max() / a[tail].mean()),
                  'max_rel_curve_gap_bf16_vs_f32_2nd_half': float(np.abs(c[tail] - a[tail]).max() / a[tail].mean()),
                  'delta_psnr_bf16_minus_f32': out['runs']['bf16']['held_out_psnr_rendered_f32'] - out['runs']['f32']['held_out_psnr_rendered_f32'],
                  'delta_psnr_f32_other_keys_minus_f32': out['runs']['f32_other_keys']['held_out_psnr_rendered_f32'] - out['runs']['f32']['held_out_psnr_rendered_f32']}
print(out['summary'])
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'bf16_convergence.json'), 'w'), indent=1)
