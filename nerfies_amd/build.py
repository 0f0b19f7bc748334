"""Builds the HIP extension in-tree (nerfies_amd/_lib/libnerfies_amd.so) for gfx950.

Each translation unit is compiled to an object in parallel (objects are cached by mtime under
nerfies_amd/_lib/obj), then linked into one shared library."""
import concurrent.futures
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB_DIR = os.path.join(HERE, '_lib')
OBJ_DIR = os.path.join(LIB_DIR, 'obj')
LIB_PATH = os.path.join(LIB_DIR, 'libnerfies_amd.so')
SOURCES = ['mlp_chain.hip', 'mlp_chain32.hip', 'mlp_bf16.hip', 'mlp_bf16x3.hip', 'warp_bf16.hip', 'warp_bf16x3.hip', 'warp_chain.hip', 'wgrad.hip', 'wgrad_bf16.hip', 'ray_kernels.hip', 'camera.hip', 'time_encoder.hip', 'nrf_plan.hip', 'nrf_run.hip', 'nrf_api.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC']
# per-source extras.  mlp_bf16.hip: every chunk of the bf16 chains is a fully unrolled `#pragma unroll` loop (34-42 MFMA slots with
# the previous panel's epilogue threaded through); above LLVM's default pragma-unroll threshold (16 k instructions) the loops
# stay rolled until after SROA and the register arrays end up in scratch.
EXTRA_FLAGS = {'mlp_bf16.hip': ['-mllvm', '-pragma-unroll-threshold=1000000'], 'mlp_bf16x3.hip': ['-mllvm', '-pragma-unroll-threshold=1000000'], 'warp_bf16x3.hip': ['-mllvm', '-pragma-unroll-threshold=1000000'], 'warp_bf16.hip': ['-mllvm', '-pragma-unroll-threshold=1000000']}


def find_hipcc():
  for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', shutil.which('hipcc')):
    if cand and os.path.exists(cand):
      return cand
  raise RuntimeError('hipcc not found (set HIPCC or install ROCm)')


def _headers():
  hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
  hs.append(os.path.join(HERE, '..', 'include', 'nerfies_amd.h'))
  return hs


def needs_build():
  if not os.path.exists(LIB_PATH):
    return True
  t = os.path.getmtime(LIB_PATH)
  deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + _headers()
  return any(os.path.getmtime(d) > t for d in deps)


def _compile(hipcc, src, verbose):
  obj = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + '.o')
  path = os.path.join(CSRC, src)
  newest = max(os.path.getmtime(p) for p in [path] + _headers())
  if os.path.exists(obj) and os.path.getmtime(obj) > newest:
    return obj
  cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ['-c', path, '-o', obj]
  if verbose:
    print(' '.join(cmd), flush=True)
  res = subprocess.run(cmd, capture_output=True, text=True)
  if res.returncode != 0:
    raise RuntimeError(f'hipcc failed on {src}:\n' + res.stderr[-4000:])
  return obj


def build(force=False, verbose=False):
  if not force and not needs_build():
    return LIB_PATH
  os.makedirs(OBJ_DIR, exist_ok=True)
  if force:
    for f in os.listdir(OBJ_DIR):
      os.remove(os.path.join(OBJ_DIR, f))
  hipcc = find_hipcc()
  with concurrent.futures.ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
    objs = list(ex.map(lambda s: _compile(hipcc, s, verbose), SOURCES))
  cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB_PATH]
  if verbose:
    print(' '.join(cmd), flush=True)
  res = subprocess.run(cmd, capture_output=True, text=True)
  if res.returncode != 0:
    raise RuntimeError('link failed:\n' + res.stderr[-4000:])
  return LIB_PATH


if __name__ == '__main__':
  import sys
  print(build(force='--force' in sys.argv, verbose=True))
