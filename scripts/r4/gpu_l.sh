#!/bin/bash
# round 4, call L: instruction-cache counters of the bf16 chain kernels; rolled trunk loops in the training kernels (the tree) against
# the fully unrolled build (variant "unrolled" = commit cd6cb7e)
O=gpurun_out/r4l; mkdir -p $O
export TMPDIR=/tmp
if [ "$1" != noic ]; then
IC="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CYCLES SQ_WAVE_CYCLES"
for m in "eval_bf16:--mode eval --bf16" "train_bf16:--mode train_bf16" "train:"; do
  tag=${m%%:*}; args=${m#*:}
  rocprofv3 --pmc $IC -d $O/pmc_ic_$tag -o pmc -- python bench.py $args --steps 3 --warmup 1 --burn-in-s 0 --no-cpu-baseline > $O/pmc_ic_$tag.log 2>&1
  f=$(find $O/pmc_ic_$tag -name '*.db' | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f $O/${tag}_pmc_icache.md; rm -rf $O/pmc_ic_$tag
  grep "mlp_\|wgrad" $O/${tag}_pmc_icache.md | grep "ICACHE\|IFETCH" | cut -c1-160
done
fi
timeout 900 python -m pytest tests/test_gpu_bf16_train.py tests/test_gpu_bf16_warp.py tests/test_gpu_round3_parity.py tests/test_gpu_bf16_convergence.py -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"
grep -h "^FAILED\|^ERROR\|^E  " $O/tests.log | head -20
for v in rolled unrolled rolled unrolled; do
  [ $v = unrolled ] && export NRF_LIB_PATH=nerfies_amd/_lib/variants/libnerfies_amd_unrolled.so || unset NRF_LIB_PATH
  for m in "train_bf16:--mode train_bf16" "fullhd_bf16:--mode fullhd --bf16" "vrig_bf16:--mode vrig --bf16" "eval_bf16:--mode eval --bf16"; do
    tag=${m%%:*}; args=${m#*:}
    timeout 300 python bench.py $args --steps 40 --warmup 5 --burn-in-s 1.5 --no-cpu-baseline > $O/ab_${v}_$tag.json 2> $O/ab.err
    python - <<PY
import json; d=json.load(open('$O/ab_${v}_$tag.json')); k=d['kernels']
print('$v $tag: %.1f k rays/s %.3f ms |' % (d['value']/1e3, d['ms_per_step']), ' '.join('%s %.3f' % (n, k[n]['ms']) for n in k if n.startswith(('mlp_','wgrad'))))
PY
  done
done
