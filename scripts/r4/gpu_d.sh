#!/bin/bash
# round 4, call D: the three tests call C left red, with their figures printed
O=gpurun_out/r4d; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_bf16_warp.py -m gpu -q -s > $O/test_gpu_bf16_warp.log 2>&1; echo "warp rc=$? $(grep -E 'passed|failed' $O/test_gpu_bf16_warp.log | tail -1)"
timeout 600 python -m pytest "tests/test_gpu_bf16_train.py::test_bf16_gradient_against_the_fp32_path" -m gpu -q -s > $O/test_grad.log 2>&1; echo "grad rc=$? $(grep -E 'passed|failed' $O/test_grad.log | tail -1)"
timeout 900 python -m pytest tests/test_gpu_bf16_convergence.py -m gpu -q -s > $O/test_conv.log 2>&1; echo "conv rc=$? $(grep -E 'passed|failed' $O/test_conv.log | tail -1)"
grep -h "^\[bf16\|^    0\.\|^    1\.\|^E " $O/*.log | head -150
