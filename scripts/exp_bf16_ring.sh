#!/bin/bash
for v in base ring160 ring96; do
  if [ $v = base ]; then unset NRF_LIB_PATH; else export NRF_LIB_PATH=$PWD/nerfies_amd/_lib/variants/libnerfies_amd_$v.so; fi
  python bench.py --mode train_bf16 --no-cpu-baseline --burn-in-s 0.5 --steps 40 > gpurun_out/bf16_ring.json 2>/dev/null
  python - $v <<'PY'
import json, sys
d = json.load(open('gpurun_out/bf16_ring.json')); k = d['kernels']
print(f"{sys.argv[1]:8s}: {d['value']/1e3:6.1f} k rays/s  wgrad_bf16 {k['wgrad_bf16']['ms']:.3f} ms")
PY
done
