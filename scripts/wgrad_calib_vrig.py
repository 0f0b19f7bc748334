"""wgrad stream-K balance and per-type costs on the gpu_vrig_paper shape (warp + elastic + background groups)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from nerfies_amd import models, training, lib as L
dev = torch.device('cuda:0')
class Cf(bench.Cfg):
  num_coarse_samples, num_fine_samples = 128, 128
  use_warp, num_warp_freqs, num_warp_features, use_camera_metadata = True, 6, 8, True
  warp_field_type = 'se3'
n = 768
model, fp = models.construct_nerf(0, Cf, n, [0,1,2,3],[0,1],[0,1,2,3], 0.0206, 0.826, device=dev)
state = training.TrainState(optimizer=training.Optimizer(fp), warp_alpha=6.0)
sp = training.ScalarParams(learning_rate=1e-3, background_loss_weight=1.0, elastic_loss_weight=0.001)
batch = bench.synthetic_batch(n, 100, dev)
g = torch.Generator().manual_seed(0)
batch['metadata'] = {'warp': torch.randint(0, 4, (n, 1), generator=g).to(dev), 'camera': torch.randint(0, 2, (n, 1), generator=g).to(dev)}
batch['background_points'] = ((torch.rand(16384, 3, generator=g) - 0.5) * 0.8).to(dev)
key = 1
for _ in range(5):
  state, stats, key = training.train_step(model, key, state, batch, sp, use_elastic_loss=True, elastic_reduce_method='weight', use_background_loss=True)
torch.cuda.synchronize()
ws = model.workspace(n, True, dev, 16384, True)
cnt = C.c_int32(0)
L.check(model.lib.nrf_debug_wgrad_segments(model.handle, None, None, C.byref(cnt)))
arr = (C.c_double * (6 * cnt.value))()
L.check(model.lib.nrf_debug_wgrad_segments(model.handle, C.c_void_p(ws.data_ptr()), arr, C.byref(cnt)))
a = np.array(arr).reshape(-1, 6)
wg, grp, Kb, Nb, nt, clk = a.T
per_wg = np.bincount(wg.astype(int), weights=clk)
print('segments', len(a), ' per-WG ticks: min %.0f mean %.0f max %.0f  (mean/max = %.3f)' % (per_wg.min(), per_wg.mean(), per_wg.max(), per_wg.mean()/per_wg.max()))
types = sorted(set(zip(Kb.astype(int), Nb.astype(int))))
X = np.zeros((len(a), len(types) + 1))
for i, (k, nb) in enumerate(zip(Kb.astype(int), Nb.astype(int))):
  X[i, types.index((k, nb))] = nt[i]
X[:, -1] = 1
coef, *_ = np.linalg.lstsq(X, clk, rcond=None)
full = coef[types.index((8, 8))]
for t, c in zip(types, coef[:-1]):
  print('type Kb=%d Nb=%d : %.2f ticks/tile  (rel %.3f)  tiles %d' % (t[0], t[1], c, c / full, int(nt[(Kb == t[0]) & (Nb == t[1])].sum())))
print('fixed per segment: %.1f ticks (rel %.3f tiles)' % (coef[-1], coef[-1] / full))
