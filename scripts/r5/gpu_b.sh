#!/bin/bash
# round 5, GPU call B: the 32-row chain kernels -- parity (A/B identity test, oracle parity files under the forced option), then the
# A/B of the headline, the 128-ray point and the eval forward at both tilings.
O=gpurun_out/r5b; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_chain32.py -x -q -m gpu > $O/pytest_chain32.log 2>&1; echo "chain32 rc=$?"; tail -5 $O/pytest_chain32.log
NRF_CHAIN_TILE_ROWS=32 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pinned.py tests/test_gpu_graph_step.py -x -q -m gpu > $O/pytest_parity32.log 2>&1; echo "parity32 rc=$?"; tail -3 $O/pytest_parity32.log
for R in 64 32; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --chain-rows $R > $O/bench_$R.json 2> $O/bench_$R.err
  timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --rays-per-gpu 128 --chain-rows $R > $O/bench128_$R.json 2> $O/bench128_$R.err
  timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --rays-per-gpu 256 --chain-rows $R > $O/bench256_$R.json 2> $O/bench256_$R.err
  timeout 300 python bench.py --mode eval --steps 20 --warmup 3 --chain-rows $R > $O/eval_$R.json 2> $O/eval_$R.err
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5b/*.json')):
  try:
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    k=d['kernels']
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), {n:(round(v['ms'],4), v['tflops'] and round(v['tflops'],1)) for n,v in k.items() if n.startswith('mlp') or n=='wgrad'})
  except Exception as e: print(f,'ERR',e)
P
