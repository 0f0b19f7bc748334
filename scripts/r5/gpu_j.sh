#!/bin/bash
# round 5, GPU call J: the warp-on bf16 convergence test with the NeRF-MLPs-only mode printed beside the full bf16 mode.
O=gpurun_out/r5j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bf16_convergence.py -q -m "gpu and not slow" -s > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
grep -h "bf16 training" $O/pytest.log | cut -c1-400
