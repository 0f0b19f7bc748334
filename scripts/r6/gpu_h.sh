#!/bin/bash
# round 6, GPU call H: embedding-gradient columns as one atomic instruction per group (bf16 SE3 reverse); fp32 wgrad per-type costs.
O=gpurun_out/r6h; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_bf16_warp.py tests/test_gpu_round3_parity.py tests/test_gpu_bf16_train.py -m gpu -q -x -p no:cacheprovider > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for m in "fullhd --bf16" "vrig --bf16"; do
  n=$(echo $m | tr -d ' -')
  timeout 300 python bench.py --mode $m --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_$n.json 2> $O/bench_$n.err
  python scripts/show_bench.py $O/bench_$n.json | head -22
done
timeout 300 python scripts/wgrad_calib.py > $O/wgrad_calib.txt 2>&1; cat $O/wgrad_calib.txt
timeout 300 python scripts/wgrad_calib_vrig.py > $O/wgrad_calib_vrig.txt 2>&1; cat $O/wgrad_calib_vrig.txt
