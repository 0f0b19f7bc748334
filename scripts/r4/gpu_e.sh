#!/bin/bash
# round 4, call E: (1) alternating DMA issuers A/B on the bf16 chains; (2) seed spread of the config-A convergence run, round-3
# bf16 kernels against the rebuilt ones; (3) the warp tests after the test-side fixes
O=gpurun_out/r4e; mkdir -p $O
export TMPDIR=/tmp
for v in split nosplit split nosplit; do
  [ $v = nosplit ] && export NRF_LIB_PATH=nerfies_amd/_lib/variants/libnerfies_amd_nosplit.so || unset NRF_LIB_PATH
  for m in "eval_bf16:--mode eval --bf16" "train_bf16:--mode train_bf16" "fullhd_bf16:--mode fullhd --bf16"; do
    tag=${m%%:*}; args=${m#*:}
    timeout 300 python bench.py $args --steps 40 --warmup 5 --burn-in-s 1.5 --no-cpu-baseline > $O/ab_${v}_$tag.json 2> $O/ab.err
    python - <<PY
import json; d=json.load(open('$O/ab_${v}_$tag.json')); k=d['kernels']
print('$v $tag: %.1f k rays/s %.3f ms |' % (d['value']/1e3, d['ms_per_step']), ' '.join('%s %.3f' % (n, k[n]['ms']) for n in k if n.startswith(('mlp_','warp_','wgrad'))))
PY
  done
done
unset NRF_LIB_PATH
timeout 600 python -m pytest tests/test_gpu_bf16_warp.py "tests/test_gpu_bf16_train.py::test_bf16_gradient_against_the_fp32_path" -m gpu -q -s > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"
grep -h "^\[bf16\|^E " $O/tests.log | head -20
SEEDS=4 MODES="bf16" NRF_LIB_PATH=nerfies_amd/_lib/variants/libnerfies_amd_r3.so timeout 600 python scripts/bf16_seed_spread.py r3kernels 2>&1 | tail -6
SEEDS=4 MODES="bf16 f32" timeout 900 python scripts/bf16_seed_spread.py r4kernels 2>&1 | tail -10
