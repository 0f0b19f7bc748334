#!/bin/bash
# round 4, call K: bf16 mode with use_alpha_condition; the bf16 suites after the G2 row / forward template change
O=gpurun_out/r4k; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_contract.py tests/test_gpu_bf16_train.py tests/test_gpu_bf16_warp.py tests/test_gpu_round3_parity.py -q -m gpu -s > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"
grep -h "^FAILED\|^ERROR\|^E  \|^\[alpha condition" $O/tests.log | head -40
timeout 300 python bench.py --mode train_bf16 --steps 40 --warmup 5 --burn-in-s 1.5 --no-cpu-baseline > $O/train_bf16.json 2>$O/b.err; python -c "
import json; d=json.load(open('$O/train_bf16.json')); k=d['kernels']; print('train_bf16 %.1f k' % (d['value']/1e3), ' '.join('%s %.3f' % (n, k[n]['ms']) for n in k if n.startswith(('mlp_','wgrad'))))"
