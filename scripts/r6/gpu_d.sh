#!/bin/bash
# round 6, GPU call D: elastic_kernel without same-line atomics (micro bench + the tests that read its sums), stream-K cost sweep of wgrad_bf16.
O=gpurun_out/r6d; mkdir -p $O
export TMPDIR=/tmp
scripts/micro/_bin/elastic_bench > $O/elastic_bench.txt 2>&1; cat $O/elastic_bench.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_onehop.py tests/test_gpu_bf16_warp.py tests/test_gpu_graph_step.py -m gpu -q -x -p no:cacheprovider > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
timeout 1500 python scripts/r6/cost_sweep.py $O/cost_sweep.json 2> $O/cost_sweep.err | tee $O/cost_sweep.txt
