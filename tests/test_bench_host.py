"""Host-side pieces of bench.py that need no GPU: the HBM-traffic table is only quoted while the kernel sources it was
measured at are unchanged; the committed table matches the committed sources; the burn-in runs whole rounds."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _fake_repo(tmp_path, kernels, sha=None):
  csrc = tmp_path / 'nerfies_amd' / 'csrc'
  csrc.mkdir(parents=True)
  (csrc / 'a.hip').write_text('__global__ void k() {}\n')
  (tmp_path / 'profiles').mkdir()
  return csrc


def test_traffic_is_null_once_the_kernels_changed(tmp_path, monkeypatch):
  csrc = _fake_repo(tmp_path, None)
  monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
  sha = bench.kernel_source_sha()
  table = {'csrc_sha16': sha, 'source': 'profiles/x.md', 'kernels': {'nrf::wgrad_kernel': {'fetch_bytes': 5e9, 'write_bytes': 1e8}}}
  (tmp_path / 'profiles' / 'hbm_traffic.json').write_text(json.dumps(table))
  assert bench.hbm_traffic('wgrad') == (5.1e9, 'profiles/x.md')
  assert bench.hbm_traffic('mlp_fwd_fine')[0] is None            # no unambiguous PMC entry for that profile name
  (csrc / 'a.hip').write_text('__global__ void k() { /* edited */ }\n')
  assert bench.kernel_source_sha() != sha
  value, why = bench.hbm_traffic('wgrad')
  assert value is None and why.startswith('stale')


def test_committed_traffic_table_matches_the_committed_kernels():
  rec = json.load(open(os.path.join(ROOT, 'profiles', 'hbm_traffic.json')))
  assert rec['csrc_sha16'] == bench.kernel_source_sha(), 'kernels changed since the PMC passes: rerun scripts/gpu_profile_round.sh'
  for sym in bench.TRAFFIC_KERNEL.values():
    assert sym in rec['kernels']
  # wgrad reads X and dY once: within 15 % of the algorithmic 5.2 GB (fp32) / 2.59 GB + the doubly-read buffers (bf16)
  rows = bench.RAYS_PER_GPU * (bench.N_COARSE + bench.N_COARSE + bench.N_FINE)
  f32 = rec['kernels']['nrf::wgrad_kernel']['fetch_bytes']
  assert 0.95 < f32 / (rows * 19.8e3) < 1.2
  b16 = rec['kernels']['nrf::wgrad_bf16_kernel']['fetch_bytes']
  assert 1.0 < b16 / (rows * 9864) < 1.2


def test_burn_in_runs_whole_rounds(monkeypatch):
  calls = []
  monkeypatch.setattr(bench.torch.cuda, 'synchronize', lambda *a, **k: None)
  n = bench.burn_in(lambda: calls.append(1), 0.0)
  assert n == 16 and len(calls) == 16
