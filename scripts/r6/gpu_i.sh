#!/bin/bash
# round 6, GPU call I: fp32 wgrad stream-K constants re-fitted: balance (calib scripts), headline / vrig / fullhd fp32 lines, parity subset.
O=gpurun_out/r6i; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python scripts/wgrad_calib.py > $O/wgrad_calib.txt 2>&1; cat $O/wgrad_calib.txt
timeout 300 python scripts/wgrad_calib_vrig.py > $O/wgrad_calib_vrig.txt 2>&1; cat $O/wgrad_calib_vrig.txt
for m in "train" "vrig" "fullhd"; do
  n=$(echo $m | tr -d ' -')
  timeout 300 python bench.py --mode $m --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_$n.json 2> $O/bench_$n.err
  python scripts/show_bench.py $O/bench_$n.json | head -22
done
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
