#!/usr/bin/env python
"""Prints a bench.py JSON line compactly: headline, roofline, per-kernel table."""
import json, sys
for path in sys.argv[1:]:
  try:
    d = json.load(open(path))
  except Exception as e:
    print(path, 'unreadable:', e); continue
  print(f"{path}: {d['value']/1e3:.1f} k rays/s, {d['ms_per_step']:.3f} ms/step, dtype {d['dtype']}, step {d.get('step_tflops', 0):.1f} TF")
  r = d['roofline']
  print(f"  roofline: {r['kernel']} {r['bound']} {r['achieved']:.1f} / {r['peak']} {r['unit']} = {r['frac']:.3f}, kernel {r['kernel_ms']:.3f} ms, traffic {r.get('traffic')}")
  ss = d.get('steady_state')
  if ss: print('  steady state:', ss)
  tot = 0.0
  for k, v in d['kernels'].items():
    tot += v['ms'] * v['launches_per_step']
    print(f"  {k:22s} {v['ms']:.4f} ms x{v['launches_per_step']:.0f}  {('%.1f TF' % v['tflops']) if v['tflops'] else ''}")
  print(f'  sum of kernels {tot:.3f} ms')
