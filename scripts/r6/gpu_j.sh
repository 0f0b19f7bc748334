#!/bin/bash
# round 6, GPU call J: four-wave workgroups for under-filled bf16 chain launches (128 / 256-ray points, tests), fp32 wgrad constants.
O=gpurun_out/r6j; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_bf16_train.py tests/test_gpu_bf16_warp.py tests/test_gpu_round3_parity.py tests/test_gpu_graph_step.py tests/test_gpu_config_sweep.py -m gpu -q -x -p no:cacheprovider > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for r in 128 256 512; do
  timeout 300 python bench.py --mode train --bf16 --rays-per-gpu $r --graph --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_trainbf16_${r}_graph.json 2> $O/err.txt
  python scripts/show_bench.py $O/bench_trainbf16_${r}_graph.json | head -16
done
timeout 300 python bench.py --mode train --bf16 --rays-per-gpu 128 --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_trainbf16_128.json 2> $O/err.txt
python scripts/show_bench.py $O/bench_trainbf16_128.json | head -3
timeout 300 python scripts/wgrad_calib_vrig.py > $O/wgrad_calib_vrig.txt 2>&1; head -3 $O/wgrad_calib_vrig.txt
timeout 300 python bench.py --mode vrig --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_vrig.json 2> $O/err.txt
python scripts/show_bench.py $O/bench_vrig.json | head -22
