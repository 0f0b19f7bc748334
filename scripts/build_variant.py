#!/usr/bin/env python
"""Experiment builds of the same ABI: scripts/build_variant.py NAME -DFOO=1 ... -> nerfies_amd/_lib/variants/libnerfies_amd_NAME.so
(select at run time with NRF_LIB_PATH).  Objects are not cached."""
import concurrent.futures
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerfies_amd import build as B


def main():
  name, extra = sys.argv[1], sys.argv[2:]
  out_dir = os.path.join(B.LIB_DIR, 'variants')
  obj_dir = os.path.join(out_dir, 'obj_' + name)
  os.makedirs(obj_dir, exist_ok=True)
  hipcc = B.find_hipcc()

  def cc(src):
    obj = os.path.join(obj_dir, os.path.splitext(src)[0] + '.o')
    r = subprocess.run([hipcc] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + extra + ['-c', os.path.join(B.CSRC, src), '-o', obj], capture_output=True, text=True)
    if r.returncode:
      raise RuntimeError(r.stderr[-3000:])
    return obj
  with concurrent.futures.ThreadPoolExecutor(max_workers=len(B.SOURCES)) as ex:
    objs = list(ex.map(cc, B.SOURCES))
  out = os.path.join(out_dir, f'libnerfies_amd_{name}.so')
  r = subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', out], capture_output=True, text=True)
  if r.returncode:
    raise RuntimeError(r.stderr[-3000:])
  print(out)


if __name__ == '__main__':
  main()
