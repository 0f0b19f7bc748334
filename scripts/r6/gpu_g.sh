#!/bin/bash
# round 6, GPU call G: fp32 wgrad with SGPR piece tables + staggered refill: parity tests, headline / vrig / fullhd lines.
O=gpurun_out/r6g; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_pinned.py tests/test_gpu_chain32.py tests/test_gpu_bf16_train.py -m gpu -q -x -p no:cacheprovider > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for m in "train" "vrig" "fullhd --bf16" "vrig --bf16" "train --bf16"; do
  n=$(echo $m | tr -d ' -')
  timeout 300 python bench.py --mode $m --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_$n.json 2> $O/bench_$n.err
  python scripts/show_bench.py $O/bench_$n.json | head -22
done
