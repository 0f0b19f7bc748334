#!/bin/bash
# round 6, GPU call C: elastic_kernel and wgrad_bf16_kernel alone (micro benches + experiment variants), the two tests that failed in B.
O=gpurun_out/r6c; mkdir -p $O
export TMPDIR=/tmp
B=scripts/micro/_bin
env | grep -i "nccl\|rccl" > $O/env_nccl.txt
for v in "" _x1 _x2 _x4 _x7; do echo "elastic_bench$v"; timeout 120 $B/elastic_bench$v; done > $O/elastic_bench.txt 2>&1
cat $O/elastic_bench.txt
for v in "" _x1 _x2; do echo "wgrad_bf16_bench$v"; timeout 300 $B/wgrad_bf16_bench$v; done > $O/wgrad_bf16_bench.txt 2>&1
cat $O/wgrad_bf16_bench.txt
timeout 900 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_contract.py -m gpu -q -x -k "rccl or warp_kwargs" -p no:cacheprovider > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
