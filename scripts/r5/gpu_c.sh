#!/bin/bash
# round 5, GPU call C: the whole -m gpu suite on the round's changes (auto tiling rule, band-parallel eval, one-hop BASELINE shapes,
# reference-side directional derivative, self-launching bench test), then the 128-ray point and the headline under the auto rule.
O=gpurun_out/r5c; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/ -x -q -m gpu -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
grep -h "one-hop\|\[2 ranks\|bf16 convergence\|bf16 training" $O/pytest_gpu.log | head -40
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --rays-per-gpu 128 > $O/bench_train128.json 2> $O/bench_train128.err
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --rays-per-gpu 128 --graph > $O/bench_train128_graph.json 2> $O/bench_train128_graph.err
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5c/*.json')):
  try:
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    k=d['kernels']
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), {n:(round(v['ms'],4), v['tflops'] and round(v['tflops'],1)) for n,v in k.items() if n.startswith('mlp') or n=='wgrad'})
  except Exception as e: print(f,'ERR',e)
P
