#!/bin/bash
# round 4, call C: bf16 SE3 trunk -- its own parity tests, the bf16 tests it touches, the warp-on bf16 bench lines
O=gpurun_out/r4c; mkdir -p $O
export TMPDIR=/tmp
for f in tests/test_gpu_bf16_warp.py tests/test_gpu_bf16.py tests/test_gpu_bf16_train.py tests/test_gpu_round3_parity.py tests/test_gpu_bf16_convergence.py; do
  n=$(basename $f .py)
  timeout 900 python -m pytest $f -m gpu -q -x -s --durations=3 > $O/$n.log 2>&1
  echo "$n rc=$? $(grep -E 'passed|failed|error' $O/$n.log | tail -1)"
done
grep -h "^\[bf16\|^E  \|assert\|Error" $O/test_gpu_bf16_warp.log | head -40
for m in "fullhd_bf16:--mode fullhd --bf16" "vrig_bf16:--mode vrig --bf16" "eval_warp_bf16:--mode eval --warp --bf16" "train_bf16:--mode train_bf16"; do
  tag=${m%%:*}; args=${m#*:}
  timeout 300 python bench.py $args --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "$tag rc=$?"
done
python scripts/show_bench.py $O/bench_fullhd_bf16.json $O/bench_vrig_bf16.json $O/bench_eval_warp_bf16.json $O/bench_train_bf16.json 2>&1 | grep -v "steady state" | tail -80
for f in $O/*.err; do echo "== $f"; tail -n 3 $f; done 2>/dev/null | tail -30
