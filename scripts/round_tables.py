#!/usr/bin/env python
"""Markdown rows + the figures the docs quote, from the bench lines of one round:  scripts/round_tables.py profiles/r03"""
import json, os, sys
pre = sys.argv[1]


def load(suf):
  f = f'{pre}_bench{suf}.json'
  if not os.path.exists(f):
    return None
  try:
    return json.loads(open(f).read().strip().splitlines()[-1])
  except Exception:
    return None


def kern(b, name):
  return b.get('kernels', {}).get(name)


def tf(b, name):
  k = kern(b, name)
  return f"{k['tflops']:.1f}" if k and k.get('tflops') else '-'


def ms(b, name):
  k = kern(b, name)
  return f"{k['ms']:.3f}" if k else '-'


rows = [('', 'train, config A, fp32 (the headline)'), ('_train_bf16', 'train, config A, bf16 mode'), ('_vrig', 'vrig shape, fp32'),
        ('_vrig_bf16', 'vrig shape, bf16 mode (NeRF MLPs + SE3 trunk; round 3: MLPs only)'), ('_fullhd', 'config D (fullhd) shape, fp32'),
        ('_fullhd_bf16', 'config D shape, bf16 mode (the precision BASELINE names)'), ('_eval', 'eval forward 8192 x (128+128), fp32, warp off'),
        ('_eval_bf16', 'eval forward, bf16 operands, warp off'),
        ('_eval_x3', 'eval forward, split-bf16 (bf16x3, float32-emulating) NeRF chains, warp off'), ('_eval_warp', 'eval forward with the SE3 warp (as eval.py renders), fp32'),
        ('_eval_warp_bf16', 'eval forward with the SE3 warp, bf16 mode'),
        ('_eval_warp_x3', 'eval forward with the SE3 warp, split-bf16 NeRF chains and SE3 trunk'),
        ('_eval_warp_x3mlp', 'same with the SE3 trunk on the float32 kernels (`--warp-f32`)'), ('_train128', 'config A, 128 rays per GPU (strong-scaling point)'),
        ('_train128_graph', 'same, whole step from one hipGraph')]
print('| line | rays/s | ms/step | step TF | roofline (dominant kernel) | traffic |')
print('|---|---|---|---|---|---|')
for suf, label in rows:
  b = load(suf)
  if not b:
    continue
  r = b['roofline']
  tr = r.get('traffic')
  trs = f"{tr / 1e9:.2f} GB" if isinstance(tr, (int, float)) else (json.dumps(tr) if tr else 'null')
  print(f"| {label} (`{os.path.basename(pre)}_bench{suf}.json`) | {b['value'] / 1e3:.1f} k | {b['ms_per_step']:.3f} | {b['step_tflops']:.1f} | "
        f"{r.get('kernel', '')} {r['achieved']:.1f} / {r['peak']:.0f} {r['unit']} = {r['frac']:.3f} | {trs} |")
for suf in ('', '_vrig', '_vrig_bf16', '_fullhd', '_fullhd_bf16', '_train_bf16', '_eval', '_eval_bf16', '_eval_x3', '_eval_warp', '_eval_warp_bf16', '_eval_warp_x3', '_eval_warp_x3mlp'):
  b = load(suf)
  if not b:
    continue
  print(f'\n{suf or "_train"}: ' + ', '.join(f"{n} {k['ms']:.3f} ms" + (f" {k['tflops']:.1f} TF" if k.get('tflops') else '') for n, k in b['kernels'].items()))
  print('   sum of kernels', b.get('sum_of_kernels_ms'), 'step/sum', b.get('step_over_sum_of_kernels'), 'cpu', b.get('cpu_baseline'))
