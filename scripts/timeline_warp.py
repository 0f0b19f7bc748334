"""Shader-clock timeline of the first tile of workgroup 0 of the SE3 chain kernels (last launch of each flavour in a step).
Needs the timeline build:  python scripts/build_variant.py timeline -DNRF_TIMELINE_BUILD
                           NRF_LIB_PATH=nerfies_amd/_lib/variants/libnerfies_amd_timeline.so python scripts/timeline_warp.py [mode]"""
import ctypes as C
import io
import os
import sys
import contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from nerfies_amd import lib as L

mode = sys.argv[1] if len(sys.argv) > 1 else 'vrig'
extra = sys.argv[2:]
sys.argv = ['bench.py', '--mode', mode, '--steps', '6', '--warmup', '3', '--burn-in-s', '0', '--no-cpu-baseline'] + extra
with contextlib.redirect_stdout(io.StringIO()):
  bench.main()
lib = L.load_library()
buf = (C.c_uint64 * (4 * 4 * 64))()
fn = lib.nrf_debug_warp_timeline
fn.argtypes = [C.c_void_p]
assert fn(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(4, 4, 64).astype(np.int64)
fwd = ['prologue', 'stash?'] 
names = {
    0: ['prologue'] + sum([[f'L{l}k', f'L{l}e'] for l in range(6)], []) + ['heads', 'se3+out'],
    1: ['prologue'] + sum([[f'L{l}k', f'L{l}e'] for l in range(6)], []) + ['heads', 'out'],
    2: ['vjp', 'headsT', 'L5k', 'L5e', 'code4', 'L4k', 'L4e', 'L3k', 'L3e', 'L2k', 'L2e', 'L1k', 'L1e', 'code0', 'scatter'],
    3: ['load', 'headsT'] + sum([[f'L{l}k', f'L{l}e'] for l in (5, 4, 3, 2, 1)], []),
}
for k, title in enumerate(['fwd primal', 'fwd tangent', 'bwd primal', 'bwd tangent']):
  print(title)
  for w in (0, 3):
    row = t[k, w]
    n = len(names[k]) + 1
    if row[0] == 0:
      print('  (not run)')
      break
    d = np.diff(row[:n])
    print(f'  wave {w}: total {row[n - 1] - row[0]}  ' + ' '.join(f'{a}={b}' for a, b in zip(names[k], d)))
