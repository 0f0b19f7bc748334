#!/bin/bash
# round 4, call A: the rebuilt bf16 chain kernels -- parity tests first, then the three bf16 bench lines
mkdir -p gpurun_out/r4a
export TMPDIR=/tmp
for f in tests/test_gpu_bf16.py tests/test_gpu_bf16_train.py tests/test_gpu_round3_parity.py; do
  n=$(basename $f .py)
  timeout 900 python -m pytest $f -m gpu -q -x --durations=5 > gpurun_out/r4a/$n.log 2>&1
  echo "$n rc=$? $(grep -E 'passed|failed|error' gpurun_out/r4a/$n.log | tail -1)"
done
timeout 300 python bench.py --mode eval --bf16 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4a/bench_eval_bf16.json 2> gpurun_out/r4a/bench_eval_bf16.err; echo "eval_bf16 rc=$?"
timeout 300 python bench.py --mode train_bf16 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r4a/bench_train_bf16.json 2> gpurun_out/r4a/bench_train_bf16.err; echo "train_bf16 rc=$?"
timeout 300 python bench.py --mode fullhd --bf16 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r4a/bench_fullhd_bf16.json 2> gpurun_out/r4a/bench_fullhd_bf16.err; echo "fullhd_bf16 rc=$?"
python scripts/show_bench.py gpurun_out/r4a/bench_eval_bf16.json gpurun_out/r4a/bench_train_bf16.json gpurun_out/r4a/bench_fullhd_bf16.json 2>&1 | tail -60
tail -3 gpurun_out/r4a/*.err
