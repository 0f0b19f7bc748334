#!/bin/bash
# round 4, call G: the Jacobian-output fix (uninitialised JacobianArgs.x_rows) + the new translation / time-encoder bf16 cases
O=gpurun_out/r4g; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_contract.py tests/test_gpu_reference_onehop.py tests/test_gpu_bf16_warp.py -q -m gpu -s > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"
grep -h "^FAILED\|^ERROR\|^E  \|^\[bf16" $O/tests.log | head -40
