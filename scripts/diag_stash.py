"""Diagnostic: decode the warp-trunk stash (fragment order) and compare with the oracle activations."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
import helpers as H
from oracle import nerfies_oracle as O
from nerfies_amd import lib as L
B = 2
spec = O.ModelSpec(num_coarse_samples=32, num_fine_samples=0, num_nerf_point_freqs=3, use_warp=True)
op = O.init_params(spec, seed=3, trained_like=True, dtype=torch.float64)
batch = O.synthetic_batch(B, seed=4, dtype=torch.float64)
model, fp = H.gpu_model(spec, op, B)
gb = H.gpu_batch(batch)
grad, stats = model.loss_and_grad(fp, gb, warp_extra={'alpha': 3.5})
torch.cuda.synchronize()
ws = model.workspace(B, True, H.DEV).cpu().numpy()
def off(name, lv=0):
  o = C.c_int64(0); L.check(model.lib.nrf_debug_ws_offset(model.handle, name.encode(), lv, C.byref(o))); return o.value
def frag_decode(buf, nfeat, ntiles):
  out = np.zeros((ntiles * 64, nfeat), np.float32)
  k = np.arange(nfeat)[None, :]; p = np.arange(64)[:, None]
  g = p >> 2; q = 2 * (g >> 2) + (g & 1); kk = (g >> 1) & 1
  idx = (((k >> 5) * 8 + q) * 64 + (k & 31) + 32 * kk) * 4 + (p & 3)
  for t in range(ntiles):
    out[t * 64:(t + 1) * 64] = buf[t * nfeat * 64:(t + 1) * nfeat * 64][idx]
  return out
rows = B * 32; nt = (rows + 63) // 64
# oracle activations of the warp trunk on the coarse samples
z, pts = O.sample_along_rays(batch['origins'], batch['directions'], 32, spec.near, spec.far, False, False, None)
wp = op['warp_field']
ids = batch['metadata']['warp'][:, None, :].expand(B, 32, 1)
emb = wp['metadata_encoder']['embed']['embedding'][ids[..., 0].long()]
inp = torch.cat([O.annealed_sinusoidal_encode(pts, 8, 3.5), emb], -1).reshape(rows, -1)
x = inp; hs = []
for i in range(6):
  if i == 4: x = torch.cat([x, inp], -1)
  x = torch.relu(O.dense(wp['trunk'][f'hidden_{i}'], x)); hs.append(x)
win = frag_decode(ws[off('w_st_win'):], 64, nt)[:rows, :inp.shape[1]]
print('st_win err', np.abs(win - inp.numpy()).max())
sth = ws[off('w_st_h'):]
for l in range(6):
  hl = frag_decode(sth[l * nt * 8192:], 128, nt)[:rows]
  print('st_h[%d] err %.2e  (max %.2e)' % (l, np.abs(hl - hs[l].numpy()).max(), hs[l].abs().max().item()))
from nerfies_amd import params as P
gt = P.tree_from_flat(grad.cpu(), model.layout)
dy = ws[off('w_dy'):]
for l in range(6):
  d = frag_decode(dy[l * nt * 8192:], 128, nt)
  b = gt['warp_field']['trunk'][f'hidden_{l}']['bias'].numpy()
  print('dy[%d]: colsum vs bias grad err %.2e (max %.2e); pad-row max %.2e' % (l, np.abs(d.sum(0) - b).max(), np.abs(b).max(), np.abs(d[rows:]).max() if d.shape[0] > rows else 0))
  # weight grad recomputed on the host from the decoded stashes
  X = win if l == 0 else frag_decode(sth[(l - 1) * nt * 8192:], 128, nt)[:rows]
  Wg = X.T.astype(np.float64) @ d[:rows].astype(np.float64)
  gk = gt['warp_field']['trunk'][f'hidden_{l}']['kernel'].numpy()
  print('      host X^T dY vs GPU kernel grad (first %d rows): %.2e (max %.2e)' % (X.shape[1], np.abs(Wg - gk[:X.shape[1]]).max(), np.abs(gk).max()))
print('--- mask consistency: dy[l] != 0 only where st_h[l] > 0')
for l in range(6):
  d = frag_decode(dy[l * nt * 8192:], 128, nt)[:rows]
  hl = frag_decode(sth[l * nt * 8192:], 128, nt)[:rows]
  bad = ((d != 0) & (hl <= 0)).sum(); nz = (d != 0).sum(); act_n = (hl > 0).sum()
  # row permutation test: does some row r of d match the mask of row r' ?
  print('dy[%d]: nonzero %d, active %d, nonzero-where-inactive %d, |d| sum %.4e' % (l, nz, act_n, bad, np.abs(d).sum()))
for l in (4, 0):
  d = frag_decode(dy[l * nt * 8192:], 128, nt)[:rows]
  hl = frag_decode(sth[l * nt * 8192:], 128, nt)[:rows]
  r, c = np.nonzero((d != 0) & (hl <= 0))
  print('dy[%d] bad (row,col,val):' % l, [(int(a), int(b), float('%.2e' % d[a, b])) for a, b in zip(r, c)][:30])
  r2, c2 = np.nonzero((d == 0) & (hl > 0))
  print('   zero-where-active:', len(r2), [(int(a), int(b)) for a, b in zip(r2, c2)][:20])
print('--- forward sign bits vs st_h > 0')
bits = ws[off('w_bits'):].view(np.uint32)
for l in range(6):
  hl = frag_decode(sth[l * nt * 8192:], 128, nt)   # [rows_pad][128]
  nbad = 0; bad = []
  for t in range(nt):
    for w in range(4):
      for lane in range(64):
        mb = int(bits[((l * nt + t) * 4 + w) * 64 + lane]); j, hh = lane & 31, lane >> 5
        for q in range(8):
          g = (q & 1) + 2 * hh + 4 * (q >> 1)
          for e in range(4):
            bit = (mb >> (4 * q + e)) & 1
            want = 1 if hl[t * 64 + 4 * g + e, 32 * w + j] > 0 else 0
            if bit != want: nbad += 1; bad.append((t * 64 + 4 * g + e, 32 * w + j, q, e, bit))
  print('layer %d: %d mismatching bits' % (l, nbad), bad[:8])
print('--- are the bad dy[4] entries the unmasked values?')
d5 = frag_decode(dy[5 * nt * 8192:], 128, nt)[:rows].astype(np.float64)
W5 = op['warp_field']['trunk']['hidden_5']['kernel'].numpy()   # [128 in, 128 out]
dh = d5 @ W5.T
d4 = frag_decode(dy[4 * nt * 8192:], 128, nt)[:rows]
h4 = frag_decode(sth[4 * nt * 8192:], 128, nt)[:rows]
r, c = np.nonzero((d4 != 0) & (h4 <= 0))
for a, b in list(zip(r, c))[:10]:
  print((int(a), int(b)), 'stash %.4e  unmasked dh %.4e  pre-act stash h %.3e' % (d4[a, b], dh[a, b], h4[a, b]))
ok = (h4 > 0)
print('active entries: max |stash - dh| = %.2e' % np.abs(d4 - dh)[ok].max())
print('--- which true dpre does each stash layer hold?')
true = {5: d5}
for l in range(4, -1, -1):
  W = op['warp_field']['trunk'][f'hidden_{l + 1}']['kernel'].numpy()[:128]
  hl = frag_decode(sth[l * nt * 8192:], 128, nt)[:rows]
  true[l] = (true[l + 1] @ W.T) * (hl > 0)
for l in range(6):
  d = frag_decode(dy[l * nt * 8192:], 128, nt)[:rows].astype(np.float64)
  errs = {m: np.abs(d - true[m]).max() / max(np.abs(true[m]).max(), 1e-30) for m in range(6)}
  print('stash dy[%d]: rel err vs true dpre_m:' % l, {m: float('%.2e' % e) for m, e in errs.items()})
print('--- structure of the dy[4] corruption')
d4 = frag_decode(dy[4 * nt * 8192:], 128, nt)[:rows].astype(np.float64)
t4 = true[4]
err = np.abs(d4 - t4) > 1e-6 * np.abs(t4).max()
print('wrong fraction overall %.3f' % err.mean())
print('by row (64):', ''.join('X' if err[r].mean() > 0.02 else '.' for r in range(64)))
print('by col (128):', ''.join('X' if err[:, c].mean() > 0.02 else '.' for c in range(128)))
# do wrong entries equal true dpre of another layer at the same position?
for m in range(6):
  same = np.isclose(d4, true[m], rtol=1e-4, atol=1e-9) & err
  print('  wrong entries equal to true dpre_%d at same pos: %d of %d' % (m, same.sum(), err.sum()))
# or equal to true dpre_4 at another row of the same column?
cnt = 0
for r, c in zip(*np.nonzero(err)):
  if np.isclose(t4[:, c], d4[r, c], rtol=1e-4, atol=1e-9).any(): cnt += 1
print('  wrong entries equal to true dpre_4 of ANOTHER ROW same col: %d' % cnt)
