#!/bin/bash
# round 6, final GPU call: the whole -m gpu suite + smoke() on the final sources, then every quoted figure of the round in one go
# (scripts/gpu_profile_round.sh r06a), the bf16 strong-scaling points and the micro benches of profiles/r06_experiments.md.
O=gpurun_out; mkdir -p $O/r6f
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 1800 python -m pytest tests/ -q -m gpu -s -p no:cacheprovider > $O/r6f/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r6f/pytest_gpu.log
grep -h "one-hop\|\[2 ranks\|bf16 convergence\|bf16 training\|graphed step\|UNPINNED\|worst leaf" $O/r6f/pytest_gpu.log > $O/r6f/parity_report.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r6f/smoke.txt 2>&1; tail -1 $O/r6f/smoke.txt
bash scripts/gpu_profile_round.sh r06a > $O/r6f/profile_round.log 2>&1
tail -5 $O/r6f/profile_round.log
for R in 128 256 512; do
  timeout 200 python bench.py --mode train_bf16 --rays-per-gpu $R --steps 200 --warmup 10 --burn-in-s 2 --no-cpu-baseline > $O/r06a_bench_train_bf16_$R.json 2>/dev/null
  timeout 200 python bench.py --mode train_bf16 --rays-per-gpu $R --steps 200 --warmup 10 --burn-in-s 2 --no-cpu-baseline --graph > $O/r06a_bench_train_bf16_${R}_graph.json 2>/dev/null
done
scripts/micro/_bin/elastic_bench > $O/r6f/elastic_bench.txt 2>&1
scripts/micro/_bin/wgrad_bf16_bench > $O/r6f/wgrad_bf16_bench.txt 2>&1
cat $O/r6f/elastic_bench.txt $O/r6f/wgrad_bf16_bench.txt
