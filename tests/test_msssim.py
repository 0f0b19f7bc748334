"""evaluation.compute_multiscale_ssim (stand-in for tf.image.ssim_multiscale, eval.py:60-62) against an independent
float64 NumPy/SciPy evaluation of the same published definition, plus the properties any SSIM has.  CPU only."""
import numpy as np
import pytest
import torch
from scipy import signal

from nerfies_amd import evaluation

W5 = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def _ref_msssim(a, b, L=1.0):
  g = np.exp(-0.5 * ((np.arange(11) - 5) / 1.5) ** 2); g /= g.sum()
  win = np.outer(g, g)
  f = lambda t: signal.correlate2d(t, win, mode='valid')
  c1, c2 = (0.01 * L) ** 2, (0.03 * L) ** 2
  per_channel = []
  for ch in range(a.shape[-1]):
    x, y, val = a[..., ch].astype(np.float64), b[..., ch].astype(np.float64), 1.0
    for i, w in enumerate(W5):
      if i:
        if x.shape[0] % 2: x, y = np.vstack([x, x[-1:]]), np.vstack([y, y[-1:]])
        if x.shape[1] % 2: x, y = np.hstack([x, x[:, -1:]]), np.hstack([y, y[:, -1:]])
        pool = lambda t: t.reshape(t.shape[0] // 2, 2, t.shape[1] // 2, 2).mean((1, 3))
        x, y = pool(x), pool(y)
      m1, m2 = f(x), f(y)
      s1, s2, s12 = f(x * x) - m1 * m1, f(y * y) - m2 * m2, f(x * y) - m1 * m2
      cs = (2 * s12 + c2) / (s1 + s2 + c2)
      term = cs.mean() if i < 4 else (cs * (2 * m1 * m2 + c1) / (m1 * m1 + m2 * m2 + c1)).mean()
      val *= max(term, 0.0) ** w
    per_channel.append(val)
  return float(np.mean(per_channel))


def _images(h, w, noise, seed=0):
  rng = np.random.default_rng(seed)
  yy, xx = np.mgrid[0:h, 0:w]
  base = np.stack([0.5 + 0.4 * np.sin(xx / 9.0) * np.cos(yy / 13.0), (xx + yy) % 64 / 64.0, rng.uniform(0, 1, (h, w))], -1)
  other = np.clip(base + noise * rng.normal(size=base.shape), 0, 1)
  return base.astype(np.float32), other.astype(np.float32)


@pytest.mark.parametrize('h,w,noise', [(176, 200, 0.05), (181, 233, 0.2), (256, 176, 0.01)])
def test_matches_independent_evaluation(h, w, noise):
  a, b = _images(h, w, noise)
  got = float(evaluation.compute_multiscale_ssim(torch.from_numpy(a), torch.from_numpy(b)))
  assert abs(got - _ref_msssim(a, b)) < 2e-5


def test_properties():
  a, b = _images(192, 192, 0.1)
  ta, tb = torch.from_numpy(a), torch.from_numpy(b)
  assert abs(float(evaluation.compute_multiscale_ssim(ta, ta)) - 1.0) < 1e-6
  s = float(evaluation.compute_multiscale_ssim(ta, tb))
  assert abs(s - float(evaluation.compute_multiscale_ssim(tb, ta))) < 1e-6 and 0 < s < 1
  _, c = _images(192, 192, 0.3)
  assert float(evaluation.compute_multiscale_ssim(ta, torch.from_numpy(c))) < s      # more noise, lower score
  with pytest.raises(ValueError):
    evaluation.compute_multiscale_ssim(ta[:100], tb[:100])
  m = evaluation.image_metrics(tb, ta)
  assert set(m) == {'mse', 'psnr', 'ssim'} and set(evaluation.image_metrics(tb[:64], ta[:64])) == {'mse', 'psnr'}
