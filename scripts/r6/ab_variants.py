#!/usr/bin/env python
"""Kernel times of one bench line under several experiment builds (scripts/build_variant.py NAME ...): python scripts/r6/ab_variants.py OUT.json
MODE... -- NAME [NAME ...];  'product' = the in-tree library."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out_path, rest = sys.argv[1], sys.argv[2:]
mode, names = rest[:rest.index('--')], rest[rest.index('--') + 1:]
res = {}
for name in names:
  env = dict(os.environ)
  if name != 'product':
    env['NRF_LIB_PATH'] = os.path.join(ROOT, 'nerfies_amd', '_lib', 'variants', f'libnerfies_amd_{name}.so')
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + mode + ['--steps', '30', '--warmup', '5', '--no-cpu-baseline', '--burn-in-s', '1'],
                     env=env, capture_output=True, text=True, cwd=ROOT)
  try:
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    res[name] = {'ms_per_step': d['ms_per_step'], 'kernels_ms': {k: round(v['ms'] * v['launches_per_step'], 4) for k, v in d['kernels'].items()}}
  except Exception as e:  # noqa: BLE001
    res[name] = {'error': f'{type(e).__name__}: {e}; ' + (r.stderr or '')[-300:]}
  print(name, json.dumps(res[name]), flush=True)
json.dump(res, open(out_path, 'w'), indent=1)
