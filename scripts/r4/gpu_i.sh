#!/bin/bash
# round 4, call I: the model without any condition (identity bottleneck), the two-rank graphed step, the graphed step through RCCL
O=gpurun_out/r4i; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_reference_onehop.py tests/test_gpu_contract.py tests/test_gpu_distributed.py tests/test_gpu_rccl.py tests/test_gpu_graph_step.py -q -m gpu -s > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"
grep -h "^FAILED\|^ERROR\|^E  \|^\[no cond\|^\[2 ranks\|one-hop nocond" $O/tests.log | head -40
