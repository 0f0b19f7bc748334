"""dPSNR gates of the bf16 training mode on the configurations it is FOR (SURVEY 8d "precision modes": the reference has no
reduced-precision switch, so parity of the mode is a held-out-PSNR / loss-curve statement against the fp32 path):

  * with the SE3 warp field ON (BASELINE configs[3], gpu_fullhd.gin, trains with the warp, warp_alpha annealing and the elastic
    regulariser): 300 Adam steps of a small deforming scene, warp_alpha advancing every step; TWO-sided against the spread two
    fp32 runs have among themselves (they differ only in their sampling keys);
  * (slow) 2000 Adam steps at the BASELINE config-A shape itself, 1024 rays x (64+128), lr 1e-3 -> 1e-4 -- the record
    profiles/r02_bf16_convergence.json, promoted to a test.
Two training runs that differ only in rounding diverge chaotically, so every gate is relative to the fp32 runs' own spread."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _scene_rgb(o, d, shift=None):
  """a smooth view-dependent target the network can fit; `shift` (N,3) deforms it per frame (what the warp field learns)"""
  x = o if shift is None else o + shift
  return torch.sigmoid(torch.stack([2.0 * torch.sin(3.0 * x[:, 0] + 2.0 * d[:, 1]), 2.0 * torch.cos(2.0 * x[:, 1] - 3.0 * d[:, 2]),
                                    1.5 * torch.sin(4.0 * x[:, 2] + d[:, 0])], -1))


def _psnr(a, b):
  return float(-10.0 * np.log10(((a - b) ** 2).mean().item()))


@pytest.mark.parametrize('fp,fw', [(6, 4), (8, 6)])   # a small shape, and the vrig preset's posenc widths (F_p = 8, F_w = 6, G = 8: ADVICE r4)
def test_bf16_training_with_the_warp_on_stays_inside_the_fp32_spread(fp, fw):
  from nerfies_amd import models, training
  B, K, NB, NID = 256, 600, 32, 4

  class Cfg:
    num_coarse_samples, num_fine_samples, num_nerf_point_freqs = 32, 32, fp
    sigma_activation, use_stratified_sampling, use_viewdirs = 'softplus', True, True
    use_warp, warp_field_type, num_warp_freqs, num_warp_features = True, 'se3', fw, 8
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
  runs = {}
  # at the preset's posenc widths a fifth run keeps the SE3 trunk in float32 (bf16='mlp', NRF_FLAG_WARP_F32): what the trunk's bf16 operands cost
  modes = (('f32', 1), ('f32b', 1001), ('f32c', 2002), ('bf16', 1)) + ((('bf16mlp', 1),) if fp >= 8 else ())
  for mode, key0 in modes:
    model, fpar = models.construct_nerf(7, Cfg, B, [0], [0], list(range(NID)), 0.05, 1.0, device=DEV)
    state = training.TrainState(optimizer=training.Optimizer(fpar))
    key, losses = key0, []
    for k in range(K):
      sp = training.ScalarParams(learning_rate=1e-3 * 0.1 ** (k / K), elastic_loss_weight=1e-3)   # exponential decay (defaults.gin)
      state = state.replace(warp_alpha=float(fw) * min(1.0, k / (0.5 * K)))   # linear schedule 0 -> F_w (warp_defaults.gin)
      i0 = (k % NB) * B
      batch = {'origins': o[i0:i0 + B], 'directions': d[i0:i0 + B], 'rgb': rgb[i0:i0 + B], 'metadata': {'warp': ids[i0:i0 + B]}}
      state, stats, key = training.train_step(model, key, state, batch, sp, use_elastic_loss=True, elastic_reduce_method='weight',
                                              bf16={'bf16': True, 'bf16mlp': 'mlp'}.get(mode, False))
      losses.append(stats['fine']['loss/rgb'])
    losses = torch.stack(losses).cpu().numpy()
    assert np.isfinite(losses).all()
    psnr = {tag: _psnr(em.apply({'params': fpar}, test, {'alpha': float(fw)}, **kw)['fine']['rgb'], rgb[n_train:])
            for tag, kw in (('f32', {}), ('bf16', dict(bf16=True)))}   # the same weights rendered by both inference modes
    runs[mode] = (psnr, losses)
  f32 = [runs[m] for m in ('f32', 'f32b', 'f32c')]
  p16, l16 = runs['bf16']
  ps = [p['f32'] for p, _ in f32]
  lo, hi = min(ps), max(ps)
  m32 = [l[-100:].mean() for _, l in f32]
  print(f'[bf16 training, warp on, F_p = {fp}, F_w = {fw}] held-out PSNR: fp32 runs {ps[0]:.3f} / {ps[1]:.3f} / {ps[2]:.3f} dB, bf16-trained {p16["f32"]:.3f} dB; bf16 '
        f'rendering of the same weights {f32[0][0]["bf16"] - ps[0]:+.3f} / {p16["bf16"] - p16["f32"]:+.3f} dB; mean loss of the last 100 '
        f'steps {m32[0]:.5f} / {m32[1]:.5f} / {m32[2]:.5f} / {l16[-100:].mean():.5f}')
  assert lo > 18.0                                                   # the scene is learnt at all
  # two-sided, relative to the spread of the three fp32 runs (which differ only in their sampling keys): first measurement
  # (300 steps, two fp32 runs 26.33 / 26.20 dB) had the bf16 run at 26.85 dB -- ABOVE both -- so the band has a floor
  band = max(0.1 + (hi - lo), 0.5)
  if fp >= 8:
    # F_p = 8 / F_w = 6 (the vrig preset's widths; measured round 5 on two boxes: fp32 24.60 .. 24.91 dB, full bf16 24.27 / 24.31 dB -- one
    # seed, 0.3 dB under the lowest fp32 run --, bf16 'mlp' 24.80 dB).  The follow-up with two seeds per mode over a 6000-step schedule
    # (scripts/r5/bf16_warp_gap.py, profiles/r05_bf16_warp_gap.json) ends at 40.34-40.39 / 40.39-40.56 / 40.15-40.39 dB (fp32 / bf16 /
    # 'mlp') with seeds of ONE mode up to 1.6 dB apart mid-schedule: no systematic cost, the 600-step figure is a draw inside the
    # scatter of a compressed schedule.  So this case gets a 1 dB floor -- it catches a broken kernel, not scatter -- and prints the
    # NeRF-MLPs-only mode beside the full one
    band = max(band, 1.0)
    pm = runs['bf16mlp'][0]['f32']
    print(f'[bf16 training, warp on, F_p = {fp}, F_w = {fw}] SE3 trunk kept in float32 (--bf16 mlp): {pm:.3f} dB')
    assert lo - band <= pm <= hi + band, (ps, pm)
  assert lo - band <= p16['f32'] <= hi + band, (ps, p16['f32'])
  assert min(m32) / (1.15 if fp < 8 else 1.3) <= l16[-100:].mean() <= (1.15 if fp < 8 else 1.3) * max(m32), (m32, l16[-100:].mean())
  for psnr, _ in runs.values():                                      # inference-mode gate with the warp on
    assert abs(psnr['bf16'] - psnr['f32']) <= 0.1


@pytest.mark.slow
def test_bf16_convergence_at_the_config_a_shape():
  """profiles/r02_bf16_convergence.json as a test (measured there: fp32 38.452 / 38.409 dB, bf16 38.418 dB; loss-curve gap over
  the second half 12.5 % bf16-vs-fp32 against 10.6 % fp32-vs-fp32)."""
  from nerfies_amd import models, training
  B, K, NB = 1024, 2000, 128

  class Cfg:
    num_coarse_samples, num_fine_samples, num_nerf_point_freqs = 64, 128, 8
    sigma_activation, use_stratified_sampling, use_viewdirs = 'softplus', True, True
  g = torch.Generator().manual_seed(0)
  o = (torch.rand(NB * B + 8192, 3, generator=g) - 0.5).to(DEV)
  d = torch.nn.functional.normalize(torch.randn(NB * B + 8192, 3, generator=g), dim=-1).to(DEV)
  rgb = _scene_rgb(o, d)
  em, _ = models.construct_nerf(7, type('E', (Cfg,), {'use_stratified_sampling': False}), 8192, [0], [0], [0], 0.05, 1.0, device=DEV)
  test = {'origins': o[NB * B:], 'directions': d[NB * B:], 'metadata': {}}
  res = {}
  for mode, key0 in (('f32', 1), ('f32b', 1001), ('bf16', 1), ('bf16b', 1001)):
    model, fp = models.construct_nerf(7, Cfg, B, [0], [0], [0], 0.05, 1.0, device=DEV)
    state = training.TrainState(optimizer=training.Optimizer(fp))
    key, curve = key0, []
    for k in range(K):
      sp = training.ScalarParams(learning_rate=1e-3 * (0.1 ** (k / K)))
      i0 = (k % NB) * B
      batch = {'origins': o[i0:i0 + B], 'directions': d[i0:i0 + B], 'rgb': rgb[i0:i0 + B], 'metadata': {}}
      state, stats, key = training.train_step(model, key, state, batch, sp, bf16=mode.startswith('bf16'))
      if (k + 1) % 20 == 0:
        curve.append(stats['fine']['loss/rgb'])
    res[mode] = (_psnr(em.apply({'params': fp}, test, {})['fine']['rgb'], rgb[NB * B:]), torch.stack(curve).cpu().numpy())
  (pa, ca), (pb, cb), (p16, c16), (p16b, _) = res['f32'], res['f32b'], res['bf16'], res['bf16b']
  tail = slice(len(ca) // 2, None)
  gap32 = np.abs(ca[tail] - cb[tail]).max() / ca[tail].mean()
  gap16 = np.abs(c16[tail] - ca[tail]).max() / ca[tail].mean()
  print(f'[bf16 convergence, config A shape, {K} steps] held-out PSNR fp32 {pa:.3f} / {pb:.3f} dB, bf16 {p16:.3f} / {p16b:.3f} dB '
        f'(means {0.5 * (p16 + p16b) - 0.5 * (pa + pb):+.3f}); loss-curve gap over the second half: bf16 vs fp32 {100 * gap16:.1f} %, '
        f'fp32 vs fp32 {100 * gap32:.1f} %')
  assert min(pa, pb) > 30.0
  # Two training runs that differ only in rounding or sampling keys diverge chaotically, so the gate is a statement about MEANS
  # against the measured seed-to-seed spread, read from the committed record (not hard-coded): profiles/r04_bf16_seed_spread.json,
  # four seeds per precision at THIS shape and step count: fp32 38.39 +- 0.19 dB, bf16 38.26 +- 0.13.  ONE-SIDED (round 4 accepted
  # +-0.70 dB around one run: a real 0.5 dB regression passed): the mean of two bf16 runs may sit at most 2.5 sigma_d below the mean
  # of two fp32 runs, sigma_d = sqrt(s32^2 / 2 + s16^2 / 2) = 0.16 dB -> 0.41 dB (false trip 0.6 %; a 0.5 dB regression is caught
  # 7 times in 10, a 0.7 dB one 96 in 100).
  import json
  import os
  rec = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'r04_bf16_seed_spread.json')))
  assert 'config-A shape' in rec['protocol'] and rec['round4_kernels']['steps'] == K, rec['protocol']
  s32, s16 = rec['round4_kernels']['std']['f32'], rec['round4_kernels']['std']['bf16']
  assert 0.05 < s32 < 0.5 and 0.05 < s16 < 0.5, (s32, s16)
  sigma_d = float(np.sqrt(0.5 * s32 ** 2 + 0.5 * s16 ** 2))
  assert abs(pa - pb) <= 4 * s32 * np.sqrt(2.0), (pa, pb)           # the fp32 pair itself is inside the measured spread
  assert 0.5 * (p16 + p16b) >= 0.5 * (pa + pb) - 2.5 * sigma_d, (pa, pb, p16, p16b, sigma_d)
  assert min(p16, p16b) >= 0.5 * (pa + pb) - 4.0 * float(np.sqrt(s16 ** 2 + 0.5 * s32 ** 2)), (pa, pb, p16, p16b)   # no single outlier run
  assert gap16 <= 2.0 * gap32 + 0.05
