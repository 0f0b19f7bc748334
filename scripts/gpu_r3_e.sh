#!/bin/bash
# round-3 GPU pass E: the whole -m gpu suite incl. the slow tests; SE3 kernel timelines; benches after the sort / composite / cond changes
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -c "import torch; x = torch.ones(1 << 20, device=\"cuda\"); print(\"sanity\", float(x.sum()), torch.cuda.get_device_name(0))" 2>&1 | tail -2
timeout 2400 python -m pytest tests -m gpu -q --maxfail=20 -rf --durations=12 > $O/r3e_pytest.log 2>&1; echo "pytest rc $?" >> $O/r3e_pytest.log
tail -30 $O/r3e_pytest.log
run() { name=$1; shift; timeout 300 python bench.py "$@" --steps 30 --warmup 5 --no-cpu-baseline > $O/r3e_bench_$name.json 2> $O/r3e_bench_$name.err; python scripts/show_bench.py $O/r3e_bench_$name.json || tail -5 $O/r3e_bench_$name.err; }
run train
run vrig --mode vrig
run fullhd_bf16 --mode fullhd --bf16
run train_bf16 --mode train_bf16
run train_bf16_graph --mode train_bf16 --graph
export NRF_LIB_PATH=$PWD/nerfies_amd/_lib/variants/libnerfies_amd_timeline.so
timeout 300 python scripts/exp_warp_timeline.py vrig > $O/r3e_timeline_vrig.txt 2>&1; tail -20 $O/r3e_timeline_vrig.txt
timeout 300 python scripts/exp_warp_timeline.py fullhd > $O/r3e_timeline_fullhd.txt 2>&1; tail -20 $O/r3e_timeline_fullhd.txt
unset NRF_LIB_PATH
