#!/bin/bash
# round 4, call H: several panels per chunk (layer 0, SE3 layers, heads, G1/G2) -- parity tests, then A/B against the build before
O=gpurun_out/r4h; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16_train.py tests/test_gpu_bf16_warp.py tests/test_gpu_round3_parity.py -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"
grep -h "^FAILED\|^ERROR\|^E  " $O/tests.log | head -20
for v in merged premerge merged premerge; do
  [ $v = premerge ] && export NRF_LIB_PATH=nerfies_amd/_lib/variants/libnerfies_amd_premerge.so || unset NRF_LIB_PATH
  for m in "eval_bf16:--mode eval --bf16" "eval_warp_bf16:--mode eval --warp --bf16" "train_bf16:--mode train_bf16" "fullhd_bf16:--mode fullhd --bf16" "vrig_bf16:--mode vrig --bf16"; do
    tag=${m%%:*}; args=${m#*:}
    timeout 300 python bench.py $args --steps 40 --warmup 5 --burn-in-s 1.5 --no-cpu-baseline > $O/ab_${v}_$tag.json 2> $O/ab.err
    python - <<PY
import json; d=json.load(open('$O/ab_${v}_$tag.json')); k=d['kernels']
print('$v $tag: %.1f k rays/s %.3f ms |' % (d['value']/1e3, d['ms_per_step']), ' '.join('%s %.3f' % (n, k[n]['ms']) for n in k if n.startswith(('mlp_','warp_','wgrad'))))
PY
  done
done
