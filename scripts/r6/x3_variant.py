#!/usr/bin/env python
"""Experiment builds of a few translation units only (the other objects are the in-tree build's):
  scripts/r6/x3_variant.py NAME [--src a.hip,b.hip] -DFOO=1 ...   (default --src mlp_bf16x3.hip)
-> nerfies_amd/_lib/variants/libnerfies_amd_NAME.so (NRF_LIB_PATH selects it; scripts/r6/ab_variants.py runs a bench line per build)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nerfies_amd import build as B
name, extra = sys.argv[1], sys.argv[2:]
srcs = ['mlp_bf16x3.hip']
if extra and extra[0] == '--src':
  srcs, extra = extra[1].split(','), extra[2:]
out_dir = os.path.join(B.LIB_DIR, 'variants')
os.makedirs(out_dir, exist_ok=True)
hipcc = B.find_hipcc()
objs = {}
for src in srcs:
  obj = os.path.join(out_dir, f'{os.path.splitext(src)[0]}_{name}.o')
  r = subprocess.run([hipcc] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + extra + ['-c', os.path.join(B.CSRC, src), '-o', obj], capture_output=True, text=True)
  assert r.returncode == 0, r.stderr[-3000:]
  objs[src] = obj
allobjs = [objs.get(s, os.path.join(B.OBJ_DIR, os.path.splitext(s)[0] + '.o')) for s in B.SOURCES]
out = os.path.join(out_dir, f'libnerfies_amd_{name}.so')
r = subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + allobjs + ['-o', out], capture_output=True, text=True)
assert r.returncode == 0, r.stderr[-3000:]
print(out)
