#!/bin/bash
# round-3 GPU pass G: SE3 reverse kernel with heads^T on MFMA and barrier-free code gradients; K-loop priority A/B
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -c "import torch; x = torch.ones(1 << 20, device=\"cuda\"); print(\"sanity\", float(x.sum()), torch.cuda.get_device_name(0))" 2>&1 | tail -2
timeout 2400 python -m pytest tests -m "gpu and not slow" -q --maxfail=20 -rf --durations=5 > $O/r3g_pytest.log 2>&1; echo "pytest rc $?" >> $O/r3g_pytest.log
tail -8 $O/r3g_pytest.log
run() { name=$1; shift; timeout 300 python bench.py "$@" --steps 30 --warmup 5 --no-cpu-baseline > $O/r3g_bench_$name.json 2> $O/r3g_bench_$name.err; python scripts/show_bench.py $O/r3g_bench_$name.json || tail -5 $O/r3g_bench_$name.err; }
run vrig --mode vrig
run train
run train_bf16 --mode train_bf16
export NRF_LIB_PATH=$PWD/nerfies_amd/_lib/variants/libnerfies_amd_prio.so
run vrig_prio --mode vrig
run train_prio
run train_bf16_prio --mode train_bf16
export NRF_LIB_PATH=$PWD/nerfies_amd/_lib/variants/libnerfies_amd_timeline.so
timeout 300 python scripts/exp_warp_timeline.py vrig > $O/r3g_timeline_vrig.txt 2>&1; tail -13 $O/r3g_timeline_vrig.txt
unset NRF_LIB_PATH
