#!/bin/bash
# round 4, call J: shallow trunks (identity layers) -- one-hop against the reference, pinned gradients
O=gpurun_out/r4j; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_reference_onehop.py tests/test_gpu_contract.py tests/test_gpu_parity.py -q -m gpu -s > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"
grep -h "^FAILED\|^ERROR\|^E  \|^\[trunk depth\|^\[no cond\|one-hop depth" $O/tests.log | head -40
