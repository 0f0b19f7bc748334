#!/bin/bash
# round-3 GPU pass C: tests (fixed thresholds), benches after the waterfall / peel / bias-prefetch changes, SE3 occupancy A/B
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m "gpu and not slow" -q --maxfail=20 -rf --durations=8 > $O/r3c_pytest.log 2>&1; echo "pytest rc $?" >> $O/r3c_pytest.log
tail -40 $O/r3c_pytest.log
run() { name=$1; shift; timeout 300 python bench.py "$@" --steps 30 --warmup 5 --no-cpu-baseline > $O/r3c_bench_$name.json 2> $O/r3c_bench_$name.err; python scripts/show_bench.py $O/r3c_bench_$name.json || tail -5 $O/r3c_bench_$name.err; }
run train
run vrig --mode vrig
run fullhd_bf16 --mode fullhd --bf16
run train_bf16 --mode train_bf16
run eval --mode eval
export NRF_LIB_PATH=$PWD/nerfies_amd/_lib/variants/libnerfies_amd_warp3.so
run vrig_warp3 --mode vrig
run fullhd_bf16_warp3 --mode fullhd --bf16
unset NRF_LIB_PATH
