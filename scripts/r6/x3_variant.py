#!/usr/bin/env python
"""Experiment builds of csrc/mlp_bf16x3.hip only (the other objects are the in-tree build's): scripts/r6/x3_variant.py NAME -DFOO=1 ...
-> nerfies_amd/_lib/variants/libnerfies_amd_NAME.so (NRF_LIB_PATH selects it; scripts/r6/ab_variants.py runs a bench line per build)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nerfies_amd import build as B
name, extra = sys.argv[1], sys.argv[2:]
out_dir = os.path.join(B.LIB_DIR, 'variants')
os.makedirs(out_dir, exist_ok=True)
hipcc = B.find_hipcc()
src = 'mlp_bf16x3.hip'
obj = os.path.join(out_dir, f'x3_{name}.o')
r = subprocess.run([hipcc] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + extra + ['-c', os.path.join(B.CSRC, src), '-o', obj], capture_output=True, text=True)
assert r.returncode == 0, r.stderr[-3000:]
objs = [os.path.join(B.OBJ_DIR, os.path.splitext(s)[0] + '.o') if s != src else obj for s in B.SOURCES]
out = os.path.join(out_dir, f'libnerfies_amd_{name}.so')
r = subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', out], capture_output=True, text=True)
assert r.returncode == 0, r.stderr[-3000:]
print(out)
