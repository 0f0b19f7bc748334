#!/usr/bin/env python
"""profiles/<round>_bench*.json are written BEFORE the PMC passes of the same profiling round exist; this fills their
`roofline.traffic` from profiles/hbm_traffic.json afterwards -- the same lookup bench.py does at run time (same workload key, same
kernel-source hash: lines whose `csrc_sha16` differs from the PMC file's are left alone).   scripts/stamp_traffic.py r03"""
import glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
rnd = sys.argv[1]
rec = json.load(open(os.path.join(ROOT, 'profiles', 'hbm_traffic.json')))
for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', f'{rnd}_bench*.json'))):
  suf = os.path.basename(f)[len(rnd) + len('_bench'):-len('.json')]
  mode = suf.lstrip('_') or 'train'
  line = json.loads(open(f).read().strip().splitlines()[-1])
  r = line['roofline']
  if r.get('traffic') is not None or line.get('csrc_sha16') != rec['csrc_sha16']:
    continue
  t, src = bench.hbm_traffic(r['kernel'], mode)
  r['traffic'], r['traffic_source'] = t, (src + ' [stamped after the PMC passes of the same round: scripts/stamp_traffic.py]') if t is not None else src
  open(f, 'w').write(json.dumps(line) + '\n')
  print(os.path.basename(f), r['kernel'], t, src)
