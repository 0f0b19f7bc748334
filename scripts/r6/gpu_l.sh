#!/bin/bash
# round 6, GPU call L: every reduce pass of a leaf in one launch (chained descriptors): gradient parity with the warp / elastic / background
# passes, the one-hop tests incl. the full-batch vectors, the bench lines.
O=gpurun_out/r6l; mkdir -p $O
export TMPDIR=/tmp
timeout 2000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_pinned.py tests/test_gpu_reference_onehop.py tests/test_gpu_bf16_warp.py tests/test_gpu_bf16_train.py tests/test_gpu_round3_parity.py tests/test_gpu_contract.py tests/test_gpu_rccl.py -m gpu -q -x -s -p no:cacheprovider > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt; grep "one-hop.*full batch" $O/pytest.txt
for m in "fullhd --bf16" "vrig --bf16" "train --bf16" "vrig" "train"; do
  n=$(echo $m | tr -d ' -')
  timeout 300 python bench.py --mode $m --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_$n.json 2> $O/bench_$n.err
  python scripts/show_bench.py $O/bench_$n.json | head -3; python scripts/show_bench.py $O/bench_$n.json | grep "grad_reduce\|sum of"
done
