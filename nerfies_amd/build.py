"""Builds the HIP extension in-tree (nerfies_amd/_lib/libnerfies_amd.so) for gfx950."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB_DIR = os.path.join(HERE, '_lib')
LIB_PATH = os.path.join(LIB_DIR, 'libnerfies_amd.so')
SOURCES = ['mlp_chain.hip', 'wgrad.hip', 'ray_kernels.hip', 'nrf_api.hip']


def find_hipcc():
  for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', shutil.which('hipcc')):
    if cand and os.path.exists(cand):
      return cand
  raise RuntimeError('hipcc not found (set HIPCC or install ROCm)')


def needs_build():
  if not os.path.exists(LIB_PATH):
    return True
  t = os.path.getmtime(LIB_PATH)
  deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
  deps.append(os.path.join(HERE, '..', 'include', 'nerfies_amd.h'))
  return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
  if not force and not needs_build():
    return LIB_PATH
  os.makedirs(LIB_DIR, exist_ok=True)
  cmd = [find_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared']
  cmd += [os.path.join(CSRC, s) for s in SOURCES]
  cmd += ['-o', LIB_PATH]
  if verbose:
    print(' '.join(cmd))
  res = subprocess.run(cmd, capture_output=True, text=True)
  if res.returncode != 0:
    raise RuntimeError('hipcc failed:\n' + res.stderr[-4000:])
  return LIB_PATH


if __name__ == '__main__':
  print(build(force=True, verbose=True))
