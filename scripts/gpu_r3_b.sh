#!/bin/bash
# round-3 GPU pass: full GPU test suite (new tests included) + bench lines of every training workload + calibration
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 -rf --durations=15 > $O/r3b_pytest.log 2>&1; echo "pytest rc $?" >> $O/r3b_pytest.log
tail -60 $O/r3b_pytest.log
run() { name=$1; shift; timeout 300 python bench.py "$@" --steps 30 --warmup 5 --no-cpu-baseline > $O/r3b_bench_$name.json 2> $O/r3b_bench_$name.err; python scripts/show_bench.py $O/r3b_bench_$name.json || tail -5 $O/r3b_bench_$name.err; }
run train
run train_graph --graph
run train128 --rays-per-gpu 128
run train128_graph --rays-per-gpu 128 --graph
run vrig --mode vrig
run fullhd --mode fullhd
run fullhd_bf16 --mode fullhd --bf16
run train_bf16 --mode train_bf16
timeout 200 python scripts/wgrad_calib.py > $O/r3b_calib.txt 2>&1; cat $O/r3b_calib.txt
timeout 200 python scripts/wgrad_calib_vrig.py > $O/r3b_calib_vrig.txt 2>&1; cat $O/r3b_calib_vrig.txt
