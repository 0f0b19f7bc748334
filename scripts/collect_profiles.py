#!/usr/bin/env python
"""gpurun_out/<tag>_* (written by scripts/gpu_profile_round.sh) -> profiles/<round>_*, and the merged profiles/hbm_traffic.json.
   collect_profiles.py r02d r02"""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
for f in glob.glob(os.path.join(ROOT, 'gpurun_out', f'{tag}_*.md')) + glob.glob(os.path.join(ROOT, 'gpurun_out', f'{tag}_bench*.json')):
  shutil.copy(f, os.path.join(ROOT, 'profiles', os.path.basename(f).replace(tag + '_', rnd + '_', 1)))
a = json.load(open(os.path.join(ROOT, 'gpurun_out', f'{tag}_hbm_traffic.json')))
b = json.load(open(os.path.join(ROOT, 'gpurun_out', f'{tag}_train_bf16_hbm_traffic.json')))
assert a['csrc_sha16'] == b['csrc_sha16']
out = {'note': a['note'],
       'source': f'profiles/{rnd}_pmc_fetch.md, profiles/{rnd}_pmc_write.md (bench.py --steps 3) and profiles/{rnd}_train_bf16_pmc_fetch.md, '
                 f'profiles/{rnd}_train_bf16_pmc_write.md (bench.py --mode train_bf16 --steps 3); scripts/gpu_profile_round.sh {tag}',
       'csrc_sha16': a['csrc_sha16'], 'kernels': {k: v for k, v in a['kernels'].items() if 'nrf::' in k}}
out['kernels'].update({k: v for k, v in b['kernels'].items() if 'nrf::' in k and 'bf16' in k})
json.dump(out, open(os.path.join(ROOT, 'profiles', 'hbm_traffic.json'), 'w'), indent=1)
print('csrc', out['csrc_sha16'], len(out['kernels']), 'kernels')
