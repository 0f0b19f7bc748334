"""Shader-clock timeline of workgroup 0 of the forward chain kernel (fine level)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['NRF_TIMELINE'] = '1'
import numpy as np, torch, bench
from nerfies_amd import models, lib as L
dev = torch.device('cuda:0')
train = len(sys.argv) > 1 and sys.argv[1] == 'train'
model, fp = models.construct_nerf(0, bench.Cfg, 1024, [0,1,2,3],[0,1],[0,1,2,3], 0.0206, 0.826, device=dev)
batch = bench.synthetic_batch(1024, 100, dev)
for _ in range(int(os.environ.get('WARM', 40))): model.apply({'params': fp}, batch, {}, rngs={'coarse': 1, 'fine': 2}, train=train)
torch.cuda.synchronize()
ws = model.workspace(1024, train, dev)
o = C.c_int64(0); L.check(model.lib.nrf_debug_ws_offset(model.handle, b'timeline', 1, C.byref(o)))
t = ws[o.value:o.value + 512].cpu().numpy().view(np.uint64).reshape(4, 64).astype(np.int64)
names = ['tile start', 'prologue'] + sum([[f'L{l} kloop', f'L{l} epi'] for l in range(8)], []) + ['alpha', 'bn', 'rgbh']
per_tile = len(names)
for w in (0, 3):
  print('wave', w)
  for tile in range(2):
    base = t[w, tile * per_tile]
    row = t[w, tile * per_tile: (tile + 1) * per_tile + 1]
    d = np.diff(row)
    print('  tile', tile, ' '.join('%s=%d' % (n.replace(' ', ''), x) for n, x in zip(names[1:] + ['logits+next'], d)))
    print('     total', row[-1] - row[0])
print('--- per-workgroup residency (fine level)')
n = 4 * 2048
rec = ws[o.value + 2 * 1024: o.value + 2 * 1024 + 2 * n].cpu().numpy().view(np.uint64).reshape(-1, 4).astype(np.int64)
nwg = int((rec[:, 1] > 0).sum())
rec = rec[:nwg]
t0 = rec[:, 0].min()
st, en = rec[:, 0] - t0, rec[:, 1] - t0
hw = rec[:, 2]; xcc = rec[:, 3] & 0xf; wall = rec[:, 3] >> 8
print('shader clock during the kernel: median %.0f MHz (min %.0f max %.0f)' % tuple(np.percentile((en - st) / np.maximum(wall, 1) * 100.0, [50, 0, 100])))
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
print('workgroups', nwg, 'kernel span', en.max(), ' start spread: p50 %d p90 %d max %d' % tuple(np.percentile(st, [50, 90, 100])))
print('durations: min %d median %d max %d' % (np.min(en - st), np.median(en - st), np.max(en - st)))
key = xcc * 10000 + se * 1000 + sh * 100 + cu
import collections
by = collections.defaultdict(list)
for b, k in enumerate(key): by[int(k)].append(b)
print('distinct (xcc,se,sh,cu) slots', len(by), ' WGs per slot histogram', collections.Counter(len(v) for v in by.values()))
for k in list(by)[:6]:
  print('  slot', k, 'blocks', by[k], 'start', [int(st[b]) for b in by[k]], 'end', [int(en[b]) for b in by[k]])
