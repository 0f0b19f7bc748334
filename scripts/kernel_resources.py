#!/usr/bin/env python
"""Register / scratch / LDS use of every kernel, read from hipcc's own metadata (the same flags as nerfies_amd/build.py):
    python scripts/kernel_resources.py [file.hip ...] > profiles/rNN_kernel_resources.md
A kernel with `.vgpr_spill_count` > 0 or `.private_segment_fixed_size` > 0 touches scratch memory."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerfies_amd import build as B


def main():
  srcs = sys.argv[1:] or B.SOURCES
  hipcc = B.find_hipcc()
  print('| file | kernel | VGPRs | AGPRs | SGPRs | VGPR spills | SGPR spills | scratch B | static LDS B |')
  print('|---|---|---|---|---|---|---|---|---|')
  for src in srcs:
    with tempfile.TemporaryDirectory() as td:
      out = os.path.join(td, 'k.s')
      cmd = [hipcc] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + ['-S', '--cuda-device-only', os.path.join(B.CSRC, src), '-o', out]
      r = subprocess.run(cmd, capture_output=True, text=True)
      if r.returncode:
        raise SystemExit(r.stderr[-2000:])
      text = open(out).read()
    for blk in re.findall(r'- \.agpr_count:.*?\.wavefront_size:', text, flags=re.S):
      g = lambda k: (re.search(rf'\.{k}:\s+(\S+)', blk) or [None, '?'])[1]
      name = subprocess.run(['c++filt', g('name')], capture_output=True, text=True).stdout.strip() or g('name')
      name = re.sub(r'\(.*', '', name)
      print(f"| {src} | `{name}` | {g('vgpr_count')} | {g('agpr_count')} | {g('sgpr_count')} | {g('vgpr_spill_count')} | "
            f"{g('sgpr_spill_count')} | {g('private_segment_fixed_size')} | {g('group_segment_fixed_size')} |")


if __name__ == '__main__':
  main()
