#!/usr/bin/env python
"""Experiment build (the product sources stay untouched): the 64-row reverse chain kernel WITHOUT its 23 per-lane bias-gradient
accumulators -- every tile adds its column sums into the workgroup's own small_part slice with fire-and-forget float atomics, as
csrc/mlp_chain32.hip does -- to see whether `nerf_mlp_bwd_kernel`'s 49 spilled VGPRs / 152 B of scratch (profiles/r05_kernel_resources.md;
VERDICT r4: "zero scratch in nerf_mlp_bwd_kernel") cost anything.

  python scripts/r5/variant_bwd_atomic.py      -> nerfies_amd/_lib/variants/libnerfies_amd_bwdatomic.so, prints the kernel's resources
  NRF_LIB_PATH=nerfies_amd/_lib/variants/libnerfies_amd_bwdatomic.so python bench.py ...

The patched copies of the two sources live under nerfies_amd/_lib/variants/src_bwdatomic/ (git-ignored)."""
import concurrent.futures
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nerfies_amd import build as B


def patch_chain(s):
  def rep(old, new, count=1):
    nonlocal s
    assert s.count(old) >= 1, old
    s = s.replace(old, new, count)
  rep('''  float (&db_trunk)[TRUNK_DEPTH][2] = C.db_trunk;
  float (&db_bn)[2] = C.db_bn;
  float& db_rgbh = C.db_rgbh;
  float (&dsum)[4] = C.dsum;''', '''  float (&dsum)[4] = C.dsum;
  float* spb = A.small_part + (size_t)blockIdx.x * SMALL_PART;   // bias column sums: atomics into the workgroup's own slice
  auto bias_add = [&](float* dst, float v, int hh) { v += __shfl_xor(v, 32); if (hh == 0) atomicAdd(dst, v); };''')
  rep('''        db_rgbh += (v4.x + v4.y) + (v4.z + v4.w);''', '''        brgbh += (v4.x + v4.y) + (v4.z + v4.w);''')
  rep('''      const __amdgpu_buffer_rsrc_t dy = make_rsrc(A.dy_rgbh + (size_t)tile * FRAG_TILE_128, FRAG_TILE_128 * 4);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int g = q_granule(q, h);
        const float4 d0''', '''      const __amdgpu_buffer_rsrc_t dy = make_rsrc(A.dy_rgbh + (size_t)tile * FRAG_TILE_128, FRAG_TILE_128 * 4);
      float brgbh = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int g = q_granule(q, h);
        const float4 d0''')
  rep('''        buf_store4(v4, dy, lane * 16, (wave * 8 + q) * 1024);
      }
    }
    __syncthreads();
    // ---- per-ray sums of dpre_rgbh''', '''        buf_store4(v4, dy, lane * 16, (wave * 8 + q) * 1024);
      }
      bias_add(spb + SP_DB_RGBH + n, brgbh, h);
    }
    __syncthreads();
    // ---- per-ray sums of dpre_rgbh''')
  rep('''        db_bn[cb] += bsum;''', '''        bias_add(spb + SP_DB_BN + n, bsum, h);''')
  rep('''#pragma unroll
      for (int q = 0; q < TRUNK_DEPTH; ++q)
        if (q == l - 1) { db_trunk[q][0] += bs[0]; db_trunk[q][1] += bs[1]; }''', '''#pragma unroll
      for (int cb = 0; cb < 2; ++cb) bias_add(spb + SP_DB_TRUNK + (l - 1) * TRUNK_W + wave * 64 + 32 * cb + j, bs[cb], h);''')
  # the flush keeps only the d-raw column sums (logit / alpha biases): the others were added tile by tile
  a = s.index('  float* sp = A.small_part + (size_t)blockIdx.x * SMALL_PART;\n#pragma unroll\n  for (int cb = 0; cb < 2; ++cb) {\n    const int n = wave * 64 + 32 * cb + j;')
  b = s.index('  __syncthreads();\n  if (tid < TILE_ROWS) {\n#pragma unroll\n    for (int c = 0; c < 4; ++c) dr[c * TILE_ROWS + tid] = dsum[c];')
  s = s[:a] + '  float* sp = A.small_part + (size_t)blockIdx.x * SMALL_PART;\n' + s[b:]
  for old in ('  const float (&db_trunk)[TRUNK_DEPTH][2] = C.db_trunk;\n', '  const float (&db_bn)[2] = C.db_bn;\n', '  const float db_rgbh = C.db_rgbh;\n'):
    rep(old, '')
  rep('  float db_trunk[TRUNK_DEPTH][2];\n  float db_bn[2];\n  float db_rgbh;\n', '')
  rep('#pragma unroll\n  for (int l = 0; l < TRUNK_DEPTH; ++l) c.db_trunk[l][0] = c.db_trunk[l][1] = 0.f;\n  c.db_bn[0] = c.db_bn[1] = 0.f;\n  c.db_rgbh = 0.f;\n', '')
  # the flush's logit / alpha stores must ADD as well?  No: they are plain stores of this workgroup's d-raw sums per level, as before.
  return s


def patch_api(s):
  old = '''    if (p.bwd32 && !warp_on && !bft) {   // the 32-row reverse chain ADDS its bias column sums into the workgroups' slices'''
  assert old in s
  new = '''    if (!bft && !(p.bwd32 && !warp_on)) {   // experiment: the 64-row reverse chain adds its bias column sums too
      int nt_all = 0;
      for (int lv = 0; lv < h->nlevels; ++lv) nt_all += p.ntiles[lv];
      const long long g64 = nt_all < 2 * h->num_cus ? nt_all : 2 * h->num_cus;
      for (int lv = 0; lv < h->nlevels; ++lv) z.add(ws + p.L[lv].small_part, g64 * SMALL_PART);
    }
'''
  return s.replace(old, new + old, 1)


def main():
  out_dir = os.path.join(B.LIB_DIR, 'variants')
  src_dir = os.path.join(out_dir, 'src_bwdatomic')
  obj_dir = os.path.join(out_dir, 'obj_bwdatomic')
  shutil.rmtree(src_dir, ignore_errors=True)
  os.makedirs(obj_dir, exist_ok=True)
  shutil.copytree(B.CSRC, src_dir)
  for name, fn in (('mlp_chain.hip', patch_chain), ('nrf_api.hip', patch_api)):
    p = os.path.join(src_dir, name)
    text = fn(open(p).read())
    open(p, 'w').write(text)
  # the sources include "../../include/nerfies_amd.h" relative to csrc: the copies get the absolute path
  for f in os.listdir(src_dir):
    q = os.path.join(src_dir, f)
    t = open(q).read()
    if '"../../include/nerfies_amd.h"' in t:
      open(q, 'w').write(t.replace('"../../include/nerfies_amd.h"', '"' + os.path.join(ROOT, 'include', 'nerfies_amd.h') + '"'))
  hipcc = B.find_hipcc()

  def cc(src):
    obj = os.path.join(obj_dir, os.path.splitext(src)[0] + '.o')
    r = subprocess.run([hipcc] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + ['-c', os.path.join(src_dir, src), '-o', obj], capture_output=True, text=True)
    if r.returncode:
      raise RuntimeError(src + '\n' + r.stderr[-3000:])
    return obj
  with concurrent.futures.ThreadPoolExecutor(max_workers=len(B.SOURCES)) as ex:
    objs = list(ex.map(cc, B.SOURCES))
  out = os.path.join(out_dir, 'libnerfies_amd_bwdatomic.so')
  r = subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', out], capture_output=True, text=True)
  if r.returncode:
    raise RuntimeError(r.stderr[-3000:])
  asm = os.path.join(obj_dir, 'mlp_chain.s')
  subprocess.run([hipcc] + B.FLAGS + ['-S', '--cuda-device-only', os.path.join(src_dir, 'mlp_chain.hip'), '-o', asm], check=True, capture_output=True)
  text = open(asm).read()
  for blk in re.findall(r'- \.agpr_count:.*?\.wavefront_size:', text, flags=re.S):
    g = lambda k: (re.search(rf'\.{k}:\s+(\S+)', blk) or [None, '?'])[1]
    if 'bwd' in g('name'):
      print(g('name'), 'vgprs', g('vgpr_count'), 'vgpr spills', g('vgpr_spill_count'), 'sgpr spills', g('sgpr_spill_count'), 'scratch', g('private_segment_fixed_size'))
  print(out)


if __name__ == '__main__':
  main()
