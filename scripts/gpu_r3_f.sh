#!/bin/bash
# round-3 GPU pass F: SE3 heads / code gradients on the MFMA pipe, scatter fast path, per-frame ids in the bench
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -c "import torch; x = torch.ones(1 << 20, device=\"cuda\"); print(\"sanity\", float(x.sum()), torch.cuda.get_device_name(0))" 2>&1 | tail -2
timeout 2400 python -m pytest tests -m "gpu and not slow" -q --maxfail=20 -rf --durations=5 > $O/r3f_pytest.log 2>&1; echo "pytest rc $?" >> $O/r3f_pytest.log
tail -12 $O/r3f_pytest.log
run() { name=$1; shift; timeout 300 python bench.py "$@" --steps 30 --warmup 5 --no-cpu-baseline > $O/r3f_bench_$name.json 2> $O/r3f_bench_$name.err; python scripts/show_bench.py $O/r3f_bench_$name.json || tail -5 $O/r3f_bench_$name.err; }
run vrig --mode vrig
run fullhd --mode fullhd
run fullhd_bf16 --mode fullhd --bf16
run train
export NRF_LIB_PATH=$PWD/nerfies_amd/_lib/variants/libnerfies_amd_timeline.so
timeout 300 python scripts/exp_warp_timeline.py vrig > $O/r3f_timeline_vrig.txt 2>&1; tail -13 $O/r3f_timeline_vrig.txt
unset NRF_LIB_PATH
