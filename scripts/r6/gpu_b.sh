#!/bin/bash
# round 6, GPU call B: the whole -m gpu suite (moved nerf_skips, warp_kwargs trunk shapes, re-run tolerance fix), no -x.
O=gpurun_out/r6b; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/pytest_gpu.txt 2>&1; tail -25 $O/pytest_gpu.txt
