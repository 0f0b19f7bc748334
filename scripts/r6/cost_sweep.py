#!/usr/bin/env python
"""Stream-K cost model of wgrad_bf16 (csrc/nrf_plan.hip bcost): sweep the per-chunk / per-accumulator-block / merged-shape terms on an
experiment build (scripts/build_variant.py exp -DNRF_EXPERIMENT reads NRF_BCOST_* from the environment) and report the kernel and
step times of the three bf16 training lines.  python scripts/r6/cost_sweep.py OUT.json"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, 'nerfies_amd', '_lib', 'variants', 'libnerfies_amd_exp.so')
SETTINGS = [  # (chunk, merged, quad, seg[, chunk of the narrow shapes])
    (12, 10, 0, 16), (12, 10, 0, 16, 8), (12, 10, 0, 16, 5), (12, 10, 0, 16, 2), (12, 6, 0, 16, 5), (12, 14, 0, 16, 5), (4, 4, 0.05, 16),
    (12, 10, 0, 16), (8, 8, 0.03, 16, 4), (12, 10, 0, 8, 5)]
if os.environ.get('COST_SWEEP_SETTINGS'):
  SETTINGS = json.loads(os.environ['COST_SWEEP_SETTINGS'])
MODES = [['--mode', 'fullhd', '--bf16'], ['--mode', 'vrig', '--bf16'], ['--mode', 'train', '--bf16']]
out = []
for st in SETTINGS:
  ch, mg, qd, sg = st[:4]
  nr = st[4] if len(st) > 4 else ch
  row = {'chunk': ch, 'merged': mg, 'quad': qd, 'seg': sg, 'narrow': nr}
  for m in MODES:
    env = dict(os.environ, NRF_LIB_PATH=LIB, NRF_BCOST_CHUNK=str(ch), NRF_BCOST_MERGED=str(mg), NRF_BCOST_QUAD=str(qd), NRF_BCOST_SEG=str(sg), NRF_BCOST_CHUNK_NARROW=str(nr))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + m + ['--steps', '30', '--warmup', '5', '--no-cpu-baseline', '--burn-in-s', '1'],
                       env=env, capture_output=True, text=True, cwd=ROOT)
    try:
      d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
      row[m[1]] = {'ms_per_step': d['ms_per_step'], 'wgrad_bf16_ms': d['kernels']['wgrad_bf16']['ms'], 'value': d['value']}
    except Exception as e:  # noqa: BLE001
      row[m[1]] = {'error': f'{type(e).__name__}: {e}; ' + (r.stderr or '')[-300:]}
  print(json.dumps(row), flush=True)
  out.append(row)
json.dump(out, open(sys.argv[1], 'w'), indent=1)
