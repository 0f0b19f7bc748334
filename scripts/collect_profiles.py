#!/usr/bin/env python
"""gpurun_out/<tag>_* (written by scripts/gpu_profile_round.sh) -> profiles/<round>_*, and the merged profiles/hbm_traffic.json:
{'csrc_sha16', 'note', 'modes': {bench mode key: {'source', 'kernels': {symbol: {fetch_bytes, write_bytes}}}}} -- one PMC table
per WORKLOAD (the same kernel moves different bytes per launch in different workloads).
   collect_profiles.py r03a r03"""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
for f in glob.glob(os.path.join(ROOT, 'gpurun_out', f'{tag}_*.md')) + glob.glob(os.path.join(ROOT, 'gpurun_out', f'{tag}_bench*.json')):
  shutil.copy(f, os.path.join(ROOT, 'profiles', os.path.basename(f).replace(tag + '_', rnd + '_', 1)))
modes, sha, note = {}, None, None
for f in sorted(glob.glob(os.path.join(ROOT, 'gpurun_out', f'{tag}*_hbm_traffic.json'))):
  base = os.path.basename(f)[len(tag):-len('_hbm_traffic.json')]
  mode = base.lstrip('_') or 'train'
  rec = json.load(open(f))
  assert sha in (None, rec['csrc_sha16']), 'PMC passes of different kernel sources'
  sha, note = rec['csrc_sha16'], rec['note']
  modes[mode] = {'source': rec['source'].replace(tag, rnd), 'kernels': {k: v for k, v in rec['kernels'].items() if 'nrf::' in k}}
json.dump({'note': note, 'csrc_sha16': sha, 'generated_by': f'scripts/gpu_profile_round.sh {tag}', 'modes': modes},
          open(os.path.join(ROOT, 'profiles', 'hbm_traffic.json'), 'w'), indent=1)
print('csrc', sha, {m: len(v['kernels']) for m, v in modes.items()})
