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
  table = {'csrc_sha16': sha, 'modes': {'train': {'source': 'profiles/x.md', 'kernels': {'nrf::wgrad_kernel': {'fetch_bytes': 5e9, 'write_bytes': 1e8}}},
                                        'vrig': {'source': 'profiles/y.md', 'kernels': {'nrf::wgrad_kernel': {'fetch_bytes': 9e9, 'write_bytes': 1e8}}}}}
  (tmp_path / 'profiles' / 'hbm_traffic.json').write_text(json.dumps(table))
  assert bench.hbm_traffic('wgrad') == (5.1e9, 'profiles/x.md')
  assert bench.hbm_traffic('wgrad', 'vrig') == (9.1e9, 'profiles/y.md')       # one table per workload
  assert bench.hbm_traffic('wgrad', 'fullhd')[0] is None
  assert bench.hbm_traffic('mlp_fwd_fine')[0] is None            # no unambiguous PMC entry for that profile name
  (csrc / 'a.hip').write_text('__global__ void k() { /* edited */ }\n')
  assert bench.kernel_source_sha() != sha
  value, why = bench.hbm_traffic('wgrad')
  assert value is None and why.startswith('stale')


def test_committed_traffic_table_matches_the_committed_kernels():
  """profiles/hbm_traffic.json is one PMC table per bench workload, stamped with the hash of the kernel sources it was measured
  at.  While the stamp matches, the figures must be the kernels' algorithmic traffic (+ slack); once a kernel has changed
  (mid-round, before the PMC passes are re-run) bench.py must report `traffic: null` with the reason instead of the stale
  figure -- either way no wrong number can be printed."""
  rec = json.load(open(os.path.join(ROOT, 'profiles', 'hbm_traffic.json')))
  modes = rec.get('modes') or {'train': {'kernels': rec['kernels']}}
  if rec['csrc_sha16'] != bench.kernel_source_sha():
    for name in bench.TRAFFIC_KERNEL:
      value, why = bench.hbm_traffic(name, 'train')
      assert value is None and ('stale' in why or 'not in' in why or 'no PMC' in why), (name, value, why)
    return
  train = modes['train']['kernels']
  assert 'nrf::wgrad_kernel' in train
  # wgrad reads X and dY once: within 20 % of the algorithmic 5.2 GB (fp32) / 2.59 GB + the doubly-read buffers (bf16)
  rows = bench.RAYS_PER_GPU * (bench.N_COARSE + bench.N_COARSE + bench.N_FINE)
  assert 0.95 < train['nrf::wgrad_kernel']['fetch_bytes'] / (rows * 19.8e3) < 1.2
  if 'train_bf16' in modes:
    b16 = modes['train_bf16']['kernels']['nrf::wgrad_bf16_kernel']['fetch_bytes']
    assert 1.0 < b16 / (rows * 9864) < 1.2
  for mode in modes:   # every committed workload answers for its dominant-kernel candidates
    for name, sym in bench.TRAFFIC_KERNEL.items():
      if sym in modes[mode]['kernels']:
        assert bench.hbm_traffic(name, mode)[0] > 0


def test_burn_in_runs_whole_rounds(monkeypatch):
  calls = []
  monkeypatch.setattr(bench.torch.cuda, 'synchronize', lambda *a, **k: None)
  n = bench.burn_in(lambda: calls.append(1), 0.0)
  assert n == 16 and len(calls) == 16


def test_gpus_n_becomes_its_own_launcher(monkeypatch):
  """`python bench.py --gpus N` outside torch.distributed.run must start N ranks itself (round 4: SystemExit rc 1).  No GPU
  here: the launch plan is checked, and that main() hands its own argument list to it."""
  argv = ['--gpus', '8', '--steps', '20', '--warmup', '5']
  cmd, env = bench.launch_plan(8, argv, ndev=8, environ={})
  assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node=8' in cmd and '--nnodes=1' in cmd
  assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and int(cmd[cmd.index('--master-port') + 1]) > 0
  assert cmd[-len(argv) - 1] == os.path.join(ROOT, 'bench.py') and cmd[-len(argv):] == argv
  assert env['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
  assert 'BENCH_SAME_DEVICE' not in env and 'BENCH_DIST_BACKEND' not in env and 'BENCH_OVERSUBSCRIBED' not in env   # real RCCL run
  # fewer devices than ranks (the one-GPU lease): same code path, ranks share the device over gloo, and the line says so
  cmd, env = bench.launch_plan(2, ['--gpus', '2'], ndev=1, environ={})
  assert env['BENCH_SAME_DEVICE'] == '1' and env['BENCH_DIST_BACKEND'] == 'gloo' and '2 ranks on 1' in env['BENCH_OVERSUBSCRIBED']
  seen = {}
  monkeypatch.delenv('WORLD_SIZE', raising=False)
  monkeypatch.setattr(bench, 'self_launch', lambda args, av: seen.update(gpus=args.gpus, argv=av) or 0)
  try:
    bench.main(argv)
    raise AssertionError('main() must exit with the launcher\'s return code')
  except SystemExit as e:
    assert e.code == 0
  assert seen == {'gpus': 8, 'argv': argv}


def test_rank_count_must_match_gpus(monkeypatch):
  monkeypatch.setenv('WORLD_SIZE', '4')
  try:
    bench.main(['--gpus', '2'])
    raise AssertionError('a launcher that started 4 ranks for --gpus 2 must be refused')
  except SystemExit as e:
    assert 'WORLD_SIZE=4' in str(e.code)


def test_rccl_preflight_summary_parses_an_init_log(tmp_path, monkeypatch):
  """What bench.py puts next to `rccl_ranks`: the transports RCCL connected its channels with, from the NCCL_DEBUG=INFO file
  of rank 0 (a canned two-rank log here; the one-rank RCCL log of a real box is parsed by tests/test_gpu_rccl.py)."""
  log = tmp_path / 'rank0.log'
  log.write_text('\n'.join([
      'box:101:101 [0] NCCL INFO NCCL_SOCKET_IFNAME set by environment to lo',
      'box:101:101 [0] NCCL INFO RCCL version : 2.26.6-HEAD:abc',
      'box:101:140 [0] NCCL INFO comm 0x55 rank 0 nranks 2 cudaDev 0 busId f4000 - Init START',
      'box:101:140 [0] NCCL INFO === System : maxBw 48.0 totalBw 336.0 === GPU/F4000 + XGMI[48.0] - GPU/E4000',
      'box:101:140 [0] NCCL INFO Channel 00/0 : 0[0] -> 1[1] via P2P/IPC comm 0x55 nRanks 02',
      'box:101:140 [0] NCCL INFO Channel 01/0 : 0[0] -> 1[1] via P2P/IPC comm 0x55 nRanks 02',
      'box:101:140 [0] NCCL INFO Channel 00/0 : 1[1] -> 0[0] via SHM/direct/direct',
      'box:101:140 [0] NCCL WARN something odd happened',
      'box:101:140 [0] NCCL INFO 16 coll channels, 16 collnet channels, 0 nvls channels, 16 p2p channels',
      'box:101:140 [0] NCCL INFO comm 0x55 rank 0 nranks 2 cudaDev 0 busId f4000 - Init COMPLETE']))
  s = bench.rccl_preflight_summary(str(log))
  assert s['connections_via'] == {'P2P/IPC': 2, 'SHM/direct/direct': 1} and s['coll_channels'] == [16] and s['nranks_seen'] == [2]
  assert s['init_complete'] and s['xgmi_mentions'] == 1 and s['warnings'] == ['something odd happened']
  assert 'RCCL version' in s['version_line'] and s['env_overrides'] == ['NCCL_SOCKET_IFNAME']
  assert bench.rccl_preflight_summary(None) is None
  miss = bench.rccl_preflight_summary(str(tmp_path / 'none.log'))   # a log that never appeared: says so, with what IS in the directory
  assert miss['log_lines'] == 0 and miss['missing'].endswith('none.log') and 'rank0.log' in miss['dir']
  # begin(): points RCCL's log of this rank at a file unless the user already asked for a console log
  for k in ('NCCL_DEBUG', 'NCCL_DEBUG_FILE', 'NCCL_DEBUG_SUBSYS'):
    monkeypatch.delenv(k, raising=False)
  path = bench.rccl_preflight_begin(3)
  assert path.endswith('rank3.log') and os.environ['NCCL_DEBUG'] == 'INFO' and os.environ['NCCL_DEBUG_FILE'] == path
  monkeypatch.delenv('NCCL_DEBUG_FILE')
  monkeypatch.setenv('NCCL_DEBUG', 'WARN')
  assert bench.rccl_preflight_begin(0) is None


def test_cpu_baseline_reports_every_candidate(monkeypatch):
  """cpu_baseline times the SAME optimizer step three ways and reports the fastest with the thread counts it used (round 5
  probed threads on 128 rays, timed one 1024-ray graph and understated the CPU by 2.3 x).  Shrunk here to 2 x 8 rays."""
  monkeypatch.setattr(bench, 'RAYS_PER_GPU', 16)
  monkeypatch.setattr(bench, 'N_COARSE', 8)
  monkeypatch.setattr(bench, 'N_FINE', 8)
  monkeypatch.setattr(bench, 'CPU_MICROBATCH', 8)
  monkeypatch.setenv('BENCH_CPU_NO_PROCS', '1')
  r = bench.cpu_baseline(0.5)
  assert r['kind'] == 'port' and r['unit'] == 'rays/s' and r['value'] > 0 and r['cores'] == r['cores_used'] <= r['cores_available']
  assert set(r['candidates']) == {'microbatch_grad_accumulation', 'full_batch_one_graph'}
  assert r['value'] == max(c['rays_per_s'] for c in r['candidates'].values()) and r['variant'] in r['candidates']
