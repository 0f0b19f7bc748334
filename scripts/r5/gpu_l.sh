#!/bin/bash
# round 5, GPU call L: A/B of the 64-row reverse chain with its bias-gradient accumulators replaced by atomics (experiment build of
# scripts/r5/variant_bwd_atomic.py: 173 VGPRs, 0 spills, 0 B scratch) against the product kernel (256 VGPRs, 49 spilled, 152 B scratch).
O=gpurun_out/r5l; mkdir -p $O
V=nerfies_amd/_lib/variants/libnerfies_amd_bwdatomic.so
NRF_LIB_PATH=$V timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/pytest_variant.log 2>&1; echo "variant parity rc=$?"; tail -2 $O/pytest_variant.log
for rep in 1 2; do
  timeout 200 python bench.py --steps 40 --warmup 5 --burn-in-s 2 --no-cpu-baseline > $O/ab_train_product_$rep.json 2>/dev/null
  NRF_LIB_PATH=$V timeout 200 python bench.py --steps 40 --warmup 5 --burn-in-s 2 --no-cpu-baseline > $O/ab_train_bwdatomic_$rep.json 2>/dev/null
done
timeout 200 python bench.py --mode vrig --steps 30 --warmup 5 --burn-in-s 2 --no-cpu-baseline > $O/ab_vrig_product_1.json 2>/dev/null
NRF_LIB_PATH=$V timeout 200 python bench.py --mode vrig --steps 30 --warmup 5 --burn-in-s 2 --no-cpu-baseline > $O/ab_vrig_bwdatomic_1.json 2>/dev/null
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5l/ab_*.json')):
  try:
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); k=d['kernels']
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), 'mlp_dgrad', round(k['mlp_dgrad']['ms'],4), round(k['mlp_dgrad']['tflops'],1), 'grad_reduce', round(k['grad_reduce']['ms'],4))
  except Exception as e: print(f,'ERR',e)
P
