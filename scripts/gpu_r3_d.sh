#!/bin/bash
# round-3 GPU pass D: tests; benches after the opaque-lane (spill) changes, coalesced prologue stash, d-posenc rewrite, 3 SE3
# workgroups per CU; SE3 kernel timelines
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m "gpu and not slow" -q --maxfail=20 -rf --durations=8 > $O/r3d_pytest.log 2>&1; echo "pytest rc $?" >> $O/r3d_pytest.log
tail -30 $O/r3d_pytest.log
run() { name=$1; shift; timeout 300 python bench.py "$@" --steps 30 --warmup 5 --no-cpu-baseline > $O/r3d_bench_$name.json 2> $O/r3d_bench_$name.err; python scripts/show_bench.py $O/r3d_bench_$name.json || tail -5 $O/r3d_bench_$name.err; }
run train
run vrig --mode vrig
run fullhd_bf16 --mode fullhd --bf16
run train_bf16 --mode train_bf16
run eval --mode eval
export NRF_LIB_PATH=$PWD/nerfies_amd/_lib/variants/libnerfies_amd_warp2.so
run vrig_warp2 --mode vrig
export NRF_LIB_PATH=$PWD/nerfies_amd/_lib/variants/libnerfies_amd_timeline.so
timeout 300 python scripts/exp_warp_timeline.py vrig > $O/r3d_timeline_vrig.txt 2>&1; cat $O/r3d_timeline_vrig.txt | tail -20
timeout 300 python scripts/exp_warp_timeline.py fullhd > $O/r3d_timeline_fullhd.txt 2>&1; cat $O/r3d_timeline_fullhd.txt | tail -20
unset NRF_LIB_PATH
