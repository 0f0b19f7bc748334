"""Does a ray's result depend on the size of the launch it rides in?  256 rays as one call vs two 128-ray calls."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import helpers as H
from oracle import nerfies_oracle as O

for warp in (False, True):
  for rows in (64, 32):
    spec = O.ModelSpec(num_coarse_samples=32, num_fine_samples=32, num_nerf_point_freqs=8, use_stratified_sampling=False, use_warp=warp,
                       use_camera_metadata=True)
    p = O.init_params(spec, seed=31, trained_like=True)
    b = H.gpu_batch(O.synthetic_batch(256, seed=5, dtype=torch.float32))
    rays = {k: v for k, v in b.items() if k != 'rgb'}
    sl = lambda r, a, c: {k: (v[a:c] if torch.is_tensor(v) else {kk: vv[a:c] for kk, vv in v.items()}) for k, v in r.items()}
    outs = []
    for parts in ([(0, 256)], [(0, 128), (128, 256)]):
      model, fp = H.gpu_model(spec, p, parts[0][1] - parts[0][0])
      model.set_chain_tile_rows(rows)
      res = [model.apply({'params': fp}, sl(rays, a, c), {'alpha': 4.0}, return_points=warp, return_weights=True) for a, c in parts]
      outs.append({lv: {k: torch.cat([r[lv][k] for r in res]) for k in res[0][lv]} for lv in res[0]})
    for lv in outs[0]:
      print(f'warp={warp} rows={rows} {lv}: ' + ', '.join(f'{k} {(outs[0][lv][k] - outs[1][lv][k]).abs().max().item():.2e}' for k in outs[0][lv]))
