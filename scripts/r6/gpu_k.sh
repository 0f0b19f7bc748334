#!/bin/bash
# round 6, GPU call K: the whole -m gpu suite on the split sources.
O=gpurun_out/r6k; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.txt 2>&1; tail -8 $O/pytest_gpu.txt
