#!/bin/bash
# round 6, GPU call E: wgrad_bf16 with per-segment SGPR piece tables (micro bench old / new), the bf16 training tests on it, stream-K cost sweep.
O=gpurun_out/r6e; mkdir -p $O
export TMPDIR=/tmp
for b in wgrad_bf16_bench_r5 wgrad_bf16_bench; do echo $b; timeout 300 scripts/micro/_bin/$b; done > $O/wgrad_bf16_bench.txt 2>&1; cat $O/wgrad_bf16_bench.txt
timeout 900 python -m pytest tests/test_gpu_bf16_train.py tests/test_gpu_bf16_warp.py tests/test_gpu_round3_parity.py -m gpu -q -x -p no:cacheprovider > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
timeout 1500 python scripts/r6/cost_sweep.py $O/cost_sweep.json 2> $O/cost_sweep.err | tee $O/cost_sweep.txt
