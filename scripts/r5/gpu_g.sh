#!/bin/bash
# round 5, GPU call G: the tests behind the one that stopped the final pass (-x): the rest of test_gpu_rccl, the one-hop reference
# tests (BASELINE shapes, reference-side directional derivative: first GPU run) and test_gpu_round3_parity.
O=gpurun_out/r5g; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_reference_onehop.py tests/test_gpu_round3_parity.py -q -m gpu -s > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
grep -h "one-hop" $O/pytest.log > $O/onehop_report.txt; cat $O/onehop_report.txt | cut -c1-250
