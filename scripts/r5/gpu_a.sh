#!/bin/bash
# round 5, GPU call A: the self-launching bench (N=2 on a one-GPU lease), the default line, the 128-ray point, and the bf16 MFMA
# ceiling under the power limit (scripts/micro/mfma_bf16_ceiling.hip) with hwmon power / clock sampled alongside.
O=gpurun_out/r5a; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --burn-in-s 0.5 > $O/bench_gpus2.json 2> $O/bench_gpus2.err; echo "rc=$?" >> $O/bench_gpus2.err )
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --rays-per-gpu 128 > $O/bench_128.json 2> $O/bench_128.err
# power / clock sampler: every hwmon card, 20 Hz, while the microbenchmark runs
( while true; do for h in /sys/class/drm/card*/device/hwmon/hwmon*; do
    p=$(cat $h/power1_average 2>/dev/null || cat $h/power1_input 2>/dev/null); f=$(cat $h/freq1_input 2>/dev/null)
    echo "$(date +%s.%N) $h $p $f"; done; sleep 0.05; done ) > $O/ceiling_hwmon.txt &
SP=$!
timeout 120 scripts/micro/_bin/mfma_bf16_ceiling > $O/mfma_bf16_ceiling.txt 2>&1
kill $SP
timeout 900 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_distributed.py -x -q -m gpu > $O/pytest_dist.log 2>&1
tail -3 $O/pytest_dist.log
cat $O/mfma_bf16_ceiling.txt
