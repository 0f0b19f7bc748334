"""C-ABI surface (CPU, no compute): the shared library loads without a GPU, exports every function
include/nerfies_amd.h declares, rejects bad input with NRF_E_* codes, and reports a parameter
layout whose leaves carry the flax paths of the reference."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'nerfies_amd.h')


@pytest.fixture(scope='module')
def lib():
  from nerfies_amd import build, lib as L
  build.build()          # hipcc cross-compiles gfx950 without a GPU; no-op when up to date
  return L.load_library()


def _declared():
  src = open(HEADER).read()
  src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
  return sorted(set(re.findall(r'\b(nrf_[a-z_0-9]+)\s*\(', src)))


def test_header_declares_the_documented_entries():
  names = _declared()
  for must in ('nrf_create', 'nrf_forward', 'nrf_backward', 'nrf_train_step_loss_grad', 'nrf_adam_step',
               'nrf_param_layout', 'nrf_workspace_bytes', 'nrf_last_error'):
    assert must in names


def test_every_declared_symbol_is_exported(lib):
  raw = C.CDLL(lib._name)
  for name in _declared():
    assert hasattr(raw, name), f'{name} declared in include/nerfies_amd.h but not exported'


def test_python_binding_covers_the_header(lib):
  from nerfies_amd import lib as L
  assert sorted(L.EXPORTS) == _declared()


def test_struct_sizes_match_header():
  """ctypes mirrors of the POD structs: field counts and sizes (all 4- or 8-byte fields, natural alignment)."""
  from nerfies_amd import lib as L
  src = open(HEADER).read()
  body = re.search(r'typedef struct nrf_model_desc \{(.*?)\} nrf_model_desc;', src, flags=re.S).group(1)
  body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
  fields = re.findall(r'\b(?:int32_t|float)\s+([a-z_0-9]+);', body)
  assert [f[0] for f in L.ModelDesc._fields_] == fields
  assert C.sizeof(L.ModelDesc) == 4 * len(fields)
  assert C.sizeof(L.Rays) == 8 + 10 * 8
  assert C.sizeof(L.TensorInfo) == 96 + 8 + 4 + 4
  # every other POD struct: same field names in the same order as the header, pointer / 4-byte / 8-byte fields only
  for cname, ct in (('nrf_rays', L.Rays), ('nrf_step_scalars', L.StepScalars), ('nrf_rand', L.Rand), ('nrf_level_out', L.LevelOut),
                    ('nrf_background', L.Background), ('nrf_elastic', L.Elastic), ('nrf_warp_reg', L.WarpReg)):
    body = re.search(r'typedef struct %s \{(.*?)\} %s;' % (cname, cname), src, flags=re.S).group(1)
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    names = re.findall(r'([a-z_0-9]+)\s*;', body)
    assert [f[0] for f in ct._fields_] == names, (cname, names)
  assert '#define NRF_NUM_STATS %d' % L.NRF_NUM_STATS in src
  assert '#define NRF_FLAG_WARP_JACOBIAN %du' % L.NRF_FLAG_WARP_JACOBIAN in src
  for name in ('NRF_FLAG_TRAIN', 'NRF_FLAG_NO_WARP', 'NRF_FLAG_BF16', 'NRF_FLAG_WARP_F32', 'NRF_FLAG_BF16X3'):   # every flag bit, header == ctypes side
    assert re.search(r'#define %s %du\b' % (name, getattr(L, name)), src), name


def test_ctypes_mirrors_match_the_compiled_header(tmp_path):
  """Every POD struct of include/nerfies_amd.h, compiled by the C compiler, against its ctypes mirror in nerfies_amd/lib.py:
  same size, same offset for every field (a stale mirror -- INTEGRATION.md's stub of round 2 lacked four nrf_rays fields and
  sized stats at 8 floats -- reads garbage pointers or overruns a buffer)."""
  import shutil
  import subprocess
  from nerfies_amd import lib as L
  cc = shutil.which('gcc') or shutil.which('cc')
  if cc is None:
    pytest.skip('no C compiler')
  pairs = [('nrf_model_desc', L.ModelDesc), ('nrf_tensor_info', L.TensorInfo), ('nrf_rays', L.Rays), ('nrf_dynamic_scalars', L.DynamicScalars),
           ('nrf_step_scalars', L.StepScalars), ('nrf_rand', L.Rand), ('nrf_level_out', L.LevelOut), ('nrf_outputs', L.Outputs),
           ('nrf_background', L.Background), ('nrf_elastic', L.Elastic), ('nrf_warp_reg', L.WarpReg), ('nrf_camera', L.CameraDesc),
           ('nrf_profile_entry', L.ProfileEntry)]
  lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', 'int main(void) {']
  for cname, ct in pairs:
    lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
    for fname, _ in ct._fields_:
      lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
  lines += ['  printf("NRF_NUM_STATS %d\\n", NRF_NUM_STATS);', '  printf("NRF_VERSION %d\\n", NRF_VERSION);', '  return 0;', '}']
  src = tmp_path / 'abi.c'
  src.write_text('\n'.join(lines))
  exe = tmp_path / 'abi'
  subprocess.run([cc, '-std=c99', '-Wall', '-Werror', str(src), '-o', str(exe)], check=True)
  got = {}
  for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines():
    *k, v = line.split()
    got[' '.join(k)] = int(v)
  for cname, ct in pairs:
    assert got[f'{cname} size'] == C.sizeof(ct), (cname, got[f'{cname} size'], C.sizeof(ct))
    for fname, _ in ct._fields_:
      assert got[f'{cname} {fname}'] == getattr(ct, fname).offset, (cname, fname)
  assert got['NRF_NUM_STATS'] == L.NRF_NUM_STATS
  assert C.sizeof(L.DynamicScalars) == 64


def test_integration_stub_matches_the_binding():
  """INTEGRATION.md's reference-side stub: its struct mirrors are executed and compared with nerfies_amd/lib.py field by
  field, and the stats buffer it allocates has NRF_NUM_STATS floats."""
  from nerfies_amd import lib as L
  md = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
  block = re.search(r'```python\n# nerfies/hip_backend.py.*?```', md, flags=re.S).group(0)
  code = block[len('```python\n'):-3]
  code = code[:code.index("_lib = C.CDLL")]          # the struct mirrors and constants; loading the .so is lib.py's job
  ns = {}
  exec(code, ns)
  assert ns['NRF_NUM_STATS'] == L.NRF_NUM_STATS and ns['NRF_FLAG_BF16'] == L.NRF_FLAG_BF16
  for name, ct in (('ModelDesc', L.ModelDesc), ('TensorInfo', L.TensorInfo), ('Rays', L.Rays), ('StepScalars', L.StepScalars), ('Rand', L.Rand)):
    mine = ns[name]
    assert [(n, C.sizeof(t)) for n, t in mine._fields_] == [(n, C.sizeof(t)) for n, t in ct._fields_], name
    assert C.sizeof(mine) == C.sizeof(ct)
  assert 'jnp.empty(NRF_NUM_STATS)' in md and 'jnp.empty(8)' not in md


def _desc(**kw):
  from nerfies_amd import lib as L
  d = L.ModelDesc(num_coarse_samples=64, num_fine_samples=128, use_viewdirs=1, near_plane=0.02, far_plane=0.8,
                  nerf_trunk_depth=8, nerf_trunk_width=256, nerf_rgb_branch_depth=1, nerf_rgb_branch_width=128,
                  nerf_skip_layer=4, use_stratified_sampling=1, num_nerf_point_freqs=8, num_nerf_viewdir_freqs=4,
                  sigma_activation=1, use_sample_at_infinity=1)
  for k, v in kw.items():
    setattr(d, k, v)
  return d


def test_create_validates_and_reports(lib):
  from nerfies_amd import lib as L
  h = C.c_void_p()
  assert lib.nrf_create(None, C.byref(h)) == -1
  d = _desc(nerf_trunk_width=320)
  assert lib.nrf_create(C.byref(d), C.byref(h)) == -3
  assert b'256' in lib.nrf_last_error()
  d = _desc(nerf_trunk_depth=9)                     # deeper than the kernels' 8 layers (shallower trunks run on identity layers)
  assert lib.nrf_create(C.byref(d), C.byref(h)) == -3
  d = _desc(nerf_trunk_depth=8, nerf_skip_layer=0)  # layer 0 reading its input twice: not built
  assert lib.nrf_create(C.byref(d), C.byref(h)) == -3
  d = _desc(use_warp=1, num_warp_freqs=4, num_warp_embeddings=2, num_warp_features=8, warp_trunk_depth=7)   # deeper than the warp kernels' 6
  assert lib.nrf_create(C.byref(d), C.byref(h)) == -3
  d = _desc(use_warp=1, num_warp_freqs=4, num_warp_embeddings=2, num_warp_features=8, warp_trunk_width=129)
  assert lib.nrf_create(C.byref(d), C.byref(h)) == -3
  for skip, depth in ((3, 8), (7, 8), (2, 6)):      # a skip at another layer: the caller's tree has the posenc rows in THAT layer
    hs = C.c_void_p()
    d = _desc(nerf_trunk_depth=depth, nerf_skip_layer=skip)
    assert lib.nrf_create(C.byref(d), C.byref(hs)) == 0, lib.nrf_last_error()
    ns = C.c_int32(0)
    assert lib.nrf_param_layout(hs, None, C.byref(ns)) == 0
    infos = (L.TensorInfo * ns.value)()
    assert lib.nrf_param_layout(hs, infos, C.byref(ns)) == 0
    shp = {t.name.decode(): (t.rows, t.cols) for t in infos}
    for i in range(depth):
      assert shp[f'nerf_mlps_fine/MLP_0/hidden_{i}/kernel'] == ((51 if i == 0 else 256) + (51 if i == skip else 0), 256), (skip, i)
    assert f'nerf_mlps_fine/MLP_0/hidden_{depth}/kernel' not in shp
    lib.nrf_destroy(hs)
  d = _desc(nerf_trunk_depth=6)                     # 6 layers, skip at 4: the caller's tree has six trunk leaves per MLP
  h6 = C.c_void_p()
  assert lib.nrf_create(C.byref(d), C.byref(h6)) == 0
  n = C.c_int32(0)
  assert lib.nrf_param_layout(h6, None, C.byref(n)) == 0
  infos = (L.TensorInfo * n.value)()
  assert lib.nrf_param_layout(h6, infos, C.byref(n)) == 0
  names = [t.name.decode() for t in infos]
  assert sum(nm.startswith('nerf_mlps_coarse/MLP_0/') and nm.endswith('kernel') for nm in names) == 6
  lib.nrf_destroy(h6)
  d = _desc(num_coarse_samples=2)
  assert lib.nrf_create(C.byref(d), C.byref(h)) == -2
  with pytest.raises(L.NrfError):
    L.check(-2, lib)


def test_layout_of_a_model_without_any_condition_has_no_bottleneck(lib):
  """use_viewdirs = 0 and no camera / appearance code: the reference's NerfMLP builds no bottleneck layer (modules.py:149-164), so
  the caller's tree has none; the rgb branch's first layer then takes the 256 trunk features alone."""
  from nerfies_amd import lib as L
  h = C.c_void_p()
  d = _desc(use_viewdirs=0)
  assert lib.nrf_create(C.byref(d), C.byref(h)) == 0, lib.nrf_last_error()
  n = C.c_int32(0)
  assert lib.nrf_param_layout(h, None, C.byref(n)) == 0
  infos = (L.TensorInfo * n.value)()
  assert lib.nrf_param_layout(h, infos, C.byref(n)) == 0
  names = {t.name.decode(): (t.rows, t.cols) for t in infos}
  assert not any('bottleneck' in k for k in names)
  assert names['nerf_mlps_coarse/MLP_1/hidden_0/kernel'] == (256, 128)
  assert names['nerf_mlps_fine/MLP_2/logit/kernel'] == (256, 1)
  cnt = C.c_int64(0)
  assert lib.nrf_param_count(h, C.byref(cnt)) == 0
  assert cnt.value >= sum(r * c for r, c in names.values())   # leaves are padded to 16 bytes
  lib.nrf_destroy(h)


def test_param_layout_uses_flax_paths(lib):
  from nerfies_amd import lib as L
  h = C.c_void_p()
  d = _desc(use_camera_metadata=1, num_camera_embeddings=2, num_camera_features=2)
  assert lib.nrf_create(C.byref(d), C.byref(h)) == 0
  n = C.c_int32(0)
  assert lib.nrf_param_layout(h, None, C.byref(n)) == 0
  infos = (L.TensorInfo * n.value)()
  assert lib.nrf_param_layout(h, infos, C.byref(n)) == 0
  names = {t.name.decode(): (t.rows, t.cols, t.offset) for t in infos}
  assert names['nerf_mlps_coarse/MLP_0/hidden_0/kernel'][:2] == (51, 256)
  assert names['nerf_mlps_coarse/MLP_0/hidden_4/kernel'][:2] == (256 + 51, 256)   # skip concat [h, posenc]
  assert names['nerf_mlps_fine/MLP_1/hidden_0/kernel'][:2] == (256 + 27 + 2, 128)  # bottleneck + viewdirs + camera
  assert names['nerf_mlps_fine/MLP_2/logit/kernel'][:2] == (256, 1)
  assert names['camera_encoder/embed/embedding'][:2] == (2, 2)
  # use_alpha_condition (modules.py:152-157, models.py:204-208): the appearance code widens the alpha head AND the rgb branch
  h2 = C.c_void_p()
  d2 = _desc(use_appearance_metadata=1, num_appearance_embeddings=4, num_appearance_features=8, use_alpha_condition=1)
  assert lib.nrf_create(C.byref(d2), C.byref(h2)) == 0
  n2 = C.c_int32(0)
  assert lib.nrf_param_layout(h2, None, C.byref(n2)) == 0
  infos2 = (L.TensorInfo * n2.value)()
  assert lib.nrf_param_layout(h2, infos2, C.byref(n2)) == 0
  names2 = {t.name.decode(): (t.rows, t.cols) for t in infos2}
  assert names2['nerf_mlps_coarse/MLP_2/logit/kernel'] == (256 + 8, 1)
  assert names2['nerf_mlps_coarse/MLP_1/hidden_0/kernel'] == (256 + 27 + 8, 128)
  assert names2['appearance_encoder/embed/embedding'] == (4, 8)
  assert lib.nrf_destroy(h2) == 0
  d3 = _desc(use_trunk_condition=1)
  assert lib.nrf_create(C.byref(d3), C.byref(h2)) == -3 and b'trunk' in lib.nrf_last_error()
  total = C.c_int64(0)
  assert lib.nrf_param_count(h, C.byref(total)) == 0
  # 589 956 parameters per NeRF MLP at P=51, R=29
  per_mlp = sum(r * c for k, (r, c, _) in names.items() if k.startswith('nerf_mlps_coarse/'))
  assert per_mlp == 589956
  assert all(off % 4 == 0 for _, _, off in names.values())
  assert lib.nrf_destroy(h) == 0


def test_narrow_model_reports_its_own_shapes(lib):
  """configs/test_vrig.gin trains a 128-wide trunk: the layout the caller sees has the model's shapes (the library
  runs it on a zero-padded image internally)."""
  from nerfies_amd import lib as L
  h = C.c_void_p()
  d = _desc(nerf_trunk_width=128, nerf_rgb_branch_width=64, use_warp=1, num_warp_freqs=8, num_warp_features=8,
            num_warp_embeddings=4)
  assert lib.nrf_create(C.byref(d), C.byref(h)) == 0
  n = C.c_int32(0)
  assert lib.nrf_param_layout(h, None, C.byref(n)) == 0
  infos = (L.TensorInfo * n.value)()
  assert lib.nrf_param_layout(h, infos, C.byref(n)) == 0
  names = {t.name.decode(): (t.rows, t.cols, t.offset) for t in infos}
  assert names['nerf_mlps_coarse/MLP_0/hidden_0/kernel'][:2] == (51, 128)
  assert names['nerf_mlps_coarse/MLP_0/hidden_4/kernel'][:2] == (128 + 51, 128)
  assert names['nerf_mlps_fine/bottleneck/kernel'][:2] == (128, 128)
  assert names['nerf_mlps_fine/MLP_1/hidden_0/kernel'][:2] == (128 + 27, 64)
  assert names['nerf_mlps_fine/MLP_1/logit/kernel'][:2] == (64, 3)
  assert names['nerf_mlps_fine/MLP_2/logit/kernel'][:2] == (128, 1)
  assert names['warp_field/trunk/hidden_4/kernel'][:2] == (128 + 59, 128)
  total = C.c_int64(0)
  assert lib.nrf_param_count(h, C.byref(total)) == 0
  last = max(names.values(), key=lambda t: t[2])
  assert total.value >= last[2] + last[0] * last[1] and total.value < 2 * sum(r * c for r, c, _ in names.values())
  offs = sorted((o, r * c) for r, c, o in names.values())
  assert all(a + na <= b for (a, na), (b, _) in zip(offs, offs[1:]))        # leaves do not overlap
  assert lib.nrf_destroy(h) == 0
