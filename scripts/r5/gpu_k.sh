#!/bin/bash
# round 5, GPU call K: the -m gpu suite once more on the final tree (as the driver runs it: one pass, -x), figures printed by the tests kept.
O=gpurun_out/r5k; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -x -q -m gpu -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
