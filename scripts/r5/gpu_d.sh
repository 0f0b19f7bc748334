#!/bin/bash
# round 5, GPU call D: the whole -m gpu suite again (band-parallel eval through host-staged gloo gathers; the bf16 wgrad with the
# skip-layer / bottleneck groups merged), the bf16 lines, and one FETCH_SIZE pass of the bf16 step (target: wgrad_bf16 <= 2.72 GB).
O=gpurun_out/r5d; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -x -q -m gpu -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
grep -h "one-hop" $O/pytest_gpu.log | head -20
timeout 300 python bench.py --mode train_bf16 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_train_bf16.json 2> $O/bench_train_bf16.err
timeout 300 python bench.py --mode fullhd --bf16 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_fullhd_bf16.json 2> $O/bench_fullhd_bf16.err
timeout 300 python bench.py --mode vrig --bf16 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_vrig_bf16.json 2> $O/bench_vrig_bf16.err
rm -rf $O/pmc2
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc2 -o pmc -- python bench.py --mode train_bf16 --steps 3 --warmup 1 --burn-in-s 0 --no-cpu-baseline > $O/pmc2.log 2>&1
f=$(find $O/pmc2 -name '*.db' | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f $O/train_bf16_pmc_fetch.md; rm -rf $O/pmc2
grep -i "wgrad_bf16\|FETCH" $O/train_bf16_pmc_fetch.md | head -5
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5d/*.json')):
  try:
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    k=d['kernels']
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), d['roofline']['kernel'], round(d['roofline']['achieved'],1), {n:(round(v['ms'],4), v['tflops'] and round(v['tflops'],1)) for n,v in k.items() if n.startswith('mlp') or n.startswith('wgrad') or n.startswith('warp')})
  except Exception as e: print(f,'ERR',e)
P
