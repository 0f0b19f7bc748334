"""Per-segment wall-clock of the wgrad kernel -> least-squares cost model (per-tile cost by group type + per-segment fixed)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from nerfies_amd import models, training, lib as L
dev = torch.device('cuda:0')
model, fp = models.construct_nerf(0, bench.Cfg, 1024, [0,1,2,3],[0,1],[0,1,2,3], 0.0206, 0.826, device=dev)
state = training.TrainState(optimizer=training.Optimizer(fp))
sp = training.ScalarParams(learning_rate=1e-3)
batch = bench.synthetic_batch(1024, 100, dev)
key = 1
for _ in range(5): state, stats, key = training.train_step(model, key, state, batch, sp)
torch.cuda.synchronize()
ws = model.workspace(1024, True, dev)
n = C.c_int32(0)
L.check(model.lib.nrf_debug_wgrad_segments(model.handle, None, None, C.byref(n)))
arr = (C.c_double * (6 * n.value))()
L.check(model.lib.nrf_debug_wgrad_segments(model.handle, C.c_void_p(ws.data_ptr()), arr, C.byref(n)))
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
for t, c in zip(types, coef[:-1]): print('type Kb=%d Nb=%d : %.2f ticks/tile  (rel %.3f)' % (t[0], t[1], c, c / full))
print('fixed per segment: %.1f ticks (rel %.3f tiles)' % (coef[-1], coef[-1] / full))
res = clk - X @ coef
print('fit residual rms %.1f ticks; worst WG total %.0f' % (np.sqrt((res**2).mean()), per_wg.max()))
