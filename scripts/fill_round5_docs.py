#!/usr/bin/env python
"""Round 5: fills the R5_* placeholders and the r05 table of DESIGN.md / README.md / BASELINE.md from profiles/r05_* and writes
profiles/r05_summary.md + profiles/r05_wgrad_bf16_merge_ab.md, so every quoted figure traces to a committed file.
  scripts/collect_profiles.py r05a r05 && scripts/stamp_traffic.py r05 && scripts/fill_round5_docs.py"""
import glob, json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pre = os.path.join(ROOT, 'profiles', 'r05')


def load(path):
  try:
    return json.loads([l for l in open(path) if l.startswith('{')][-1])
  except Exception:
    return None


bench = lambda suf: load(f'{pre}_bench{suf}.json')
t, t128, t128g, bf = bench(''), bench('_train128'), bench('_train128_graph'), bench('_train_bf16')
v = {}
v['R5_TRAINK'] = f"{t['value'] / 1e3:.1f}"
v['R5_T128MS'] = f"{t128['ms_per_step']:.3f}"
v['R5_T128K'] = f"{t128['value'] / 1e3:.1f}"
v['R5_T128PCT'] = f"{100 * t128['value'] / t['value']:.0f}"
# ---- the parity report: figures the -m gpu tests printed on the final sources
rep = open(pre + '_parity_report.txt').read() if os.path.exists(pre + '_parity_report.txt') else ''
dirs = re.findall(r'one-hop directional derivative (\w+): max relative .*? ([0-9.e+-]+)\s+\(slopes', rep)
v['R5_DIRERR'] = ', '.join(f'{e} ({n})' for n, e in dirs) or 'see profiles/r05_parity_report.txt'
hops = re.findall(r'one-hop (cfg\w) \(.*?\): max \|hip - reference\| (.*)', rep)
v['R5_ONEHOP'] = '; '.join(f"{n}: {', '.join(x for x in row.split(', ') if x.split()[0] in ('rgb', 'depth', 'acc'))}" for n, row in hops) or 'see profiles/r05_parity_report.txt'
# ---- A/B of the merged bf16 wgrad groups (same box, back to back, two repetitions each)
ab = {}
for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r05_ab_*.json'))):
  m = re.match(r'r05_ab_(train_bf16|fullhd_bf16)_(on|off)_(\d)\.json', os.path.basename(f))
  d = load(f)
  if m and d:
    ab.setdefault((m.group(1), m.group(2)), []).append((d['value'], d['ms_per_step'], d['kernels']['wgrad_bf16']['ms']))
mean = lambda xs: sum(xs) / len(xs)
if ab:
  v['R5_MERGEMS'] = f"{mean([x[2] for x in ab[('train_bf16', 'on')]]):.3f}"
  v['R5_NOMERGEMS'] = f"{mean([x[2] for x in ab[('train_bf16', 'off')]]):.3f}"
fetch = None
fm = os.path.join(ROOT, 'profiles', 'r05_train_bf16_pmc_fetch.md')   # the default (merge on) pass of the profile round
if os.path.exists(fm):
  for l in open(fm):
    if 'wgrad_bf16' in l and 'FETCH_SIZE' in l:
      fetch = float(l.split('|')[-2])          # average FETCH_SIZE per dispatch
if fetch:
  # guide: FETCH_SIZE counts 32-byte... units are KiB-like per the round-2 calibration: bytes = value * 1024 * 2 (gfx950 wide-load correction)
  gb = fetch * 1024 * 2 / 1e9
  rows = 1024 * (64 + 64 + 128)
  alg = rows * 2 * ((64 + 8 * 256 + 256 + 128) + (8 * 256 + 256 + 128 + 4)) / 1e9
  v['R5_MERGEGB'] = f'{gb:.2f}'
  v['R5_MERGEX'] = f'{gb / alg:.2f}'
  lines = ['# Round 5 A/B: `NRF_OPT_BF16_WGRAD_MERGE` (bf16 wgrad, skip-layer and bottleneck + alpha groups merged so that dpre_4 / h8 are streamed once)', '',
           'Same box, back to back, two repetitions each (first pass of `scripts/r5/gpu_final.sh`, when the option still defaulted to off: `bench.py --mode ... [--bf16-wgrad-merge]`, 2 s burn-in, 50 steps; files `profiles/r05_ab_*.json`).', '',
           '| workload | merge | rays/s | ms/step | wgrad_bf16 ms |', '|---|---|---|---|---|']
  for (w, m), xs in sorted(ab.items()):
    for x in xs:
      lines.append(f'| {w} | {m} | {x[0]:.0f} | {x[1]:.4f} | {x[2]:.4f} |')
  lines += ['', f'HBM fetch of `wgrad_bf16_kernel` with the merge ON (`profiles/r05_train_bf16_pmc_fetch.md`; the A/B box: `profiles/r05_ab_train_bf16_merged_pmc_fetch.md`; FETCH_SIZE x 2 per the gfx950 correction): '
            f'{gb:.2f} GB per launch = {gb / alg:.2f} x the algorithmic {alg:.2f} GB (merge off: 2.95 GB = 1.16 x in round 4, `profiles/r04_train_bf16_pmc_fetch.md`).',
            '', 'Earlier in the round (other boxes): byte-proportional stream-K cost 0.61 ms (`gpurun_out/r5d`), cost of the merged shapes swept +0 / 6 / 10 / 16 / 24 / 32 block units: '
            '0.555 / 0.508 / 0.500 / 0.520 / 0.527 / 0.543 ms (`scripts/r5/gpu_e.sh`).  Reading: 10 % fewer bytes buy 2-5 % of the kernel (1-2 % of the step): the kernel is bound '
            'mostly by its per-chunk cycle, not by HBM bytes.  Box-to-box spread of this kernel is larger than the effect (0.50 ms on one box, 0.57 on another, merge off), '
            'which is why only the same-box pairs above decide; the option is ON by default since.', '']
  open(pre + '_wgrad_bf16_merge_ab.md', 'w').write('\n'.join(lines))
for k in ('R5_MERGEMS', 'R5_NOMERGEMS', 'R5_MERGEGB', 'R5_MERGEX'):
  v.setdefault(k, 'n/a')
table = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'round_tables.py'), pre], capture_output=True, text=True).stdout
md_rows = '\n'.join(l for l in table.splitlines() if l.startswith('|'))
for name in ('DESIGN.md', 'README.md', 'BASELINE.md'):
  p = os.path.join(ROOT, name)
  s = open(p).read()
  for k, x in v.items():
    s = s.replace(k, x)
  s = re.sub(r'(<!-- r05-table[^>]*-->\n).*?(\n<!-- /r05-table -->)', lambda m: m.group(1) + md_rows + m.group(2), s, flags=re.S)
  open(p, 'w').write(s)
shas = sorted({f"{load(f).get('csrc_sha16')} ({os.path.basename(f)})" for f in glob.glob(pre + '_bench*.json') if load(f)})
open(pre + '_summary.md', 'w').write(
    '# Round 5: one-page view of `profiles/r05_*`\n\nRegenerated by `scripts/r5/gpu_final.sh` (= the -m gpu suite, the merge A/B, `scripts/gpu_profile_round.sh r05a`) + '
    '`scripts/collect_profiles.py r05a r05` + `scripts/stamp_traffic.py r05` on one MI355X; every bench line after a 3 s burn-in, per-kernel times from HIP events of the '
    'same run, `traffic` from separate `rocprofv3 --pmc` passes (`profiles/hbm_traffic.json`, stamped with the kernel-source hash).  Kernel resources: '
    '`profiles/r05_kernel_resources.md`.  Other records of the round: `r05_chain32_ab.md` (64- vs 32-row tiling), `r05_mfma_bf16_ceiling.txt`, `r05_wgrad_bf16_merge_ab.md`, '
    '`r05_bench_gpus2_one_gpu_lease.json`, `r05_parity_report.txt`.  csrc hashes of the lines: ' + ', '.join(shas) + '.\n\n' + md_rows +
    '\n\nCPU baseline (config A, torch-CPU fp32 oracle on the GPU box\'s host): ' + json.dumps(t.get('cpu_baseline')) + '\n\nPer-kernel times:\n\n```\n' +
    '\n'.join(l for l in table.splitlines() if not l.startswith('|')) + '\n```\n')
print(json.dumps(v, indent=1))
