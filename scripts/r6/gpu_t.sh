#!/bin/bash
# round 6, GPU call T: epilogue units as single volatile asm statements (x3 chain; bf16 inference chains of the NeRF MLP and the SE3 trunk):
# parity tests of the touched kernels, same-box A/B against the build without them ('old')
O=gpurun_out/r6t; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16x3.py tests/test_gpu_bf16.py tests/test_gpu_bf16_warp.py tests/test_gpu_round3_parity.py -m gpu -q -x -p no:cacheprovider > $O/tests.txt 2>&1; tail -3 $O/tests.txt
python scripts/r6/ab_variants.py $O/ab_x3.json --mode eval --split-bf16 -- product old product old 2>&1 | cut -c1-260
python scripts/r6/ab_variants.py $O/ab_bf16.json --mode eval --bf16 -- product old product old 2>&1 | cut -c1-260
python scripts/r6/ab_variants.py $O/ab_warp_bf16.json --mode eval --warp --bf16 -- product old product old 2>&1 | cut -c1-330
