#!/bin/bash
# round 5, final GPU call: the whole -m gpu suite on the final sources, then every quoted figure of the round in one go
# (scripts/gpu_profile_round.sh r05a).
O=gpurun_out; mkdir -p $O/r5f
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -x -q -m gpu -s > $O/r5f/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r5f/pytest_gpu.log
grep -h "one-hop\|\[2 ranks\|bf16 convergence\|bf16 training\|graphed step" $O/r5f/pytest_gpu.log > $O/r5f/parity_report.txt
# (the same-box A/B of NRF_OPT_BF16_WGRAD_MERGE ran in the first pass of this script, before the option became the default:
# profiles/r05_ab_*.json, two repetitions each of `bench.py --mode train_bf16 / fullhd --bf16` with and without the merge)
bash scripts/gpu_profile_round.sh r05a > $O/r5f/profile_round.log 2>&1
tail -5 $O/r5f/profile_round.log
