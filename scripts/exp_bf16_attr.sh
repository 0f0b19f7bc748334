#!/bin/bash
for v in base nostore nobits noboth; do
  if [ $v = base ]; then unset NRF_LIB_PATH; else export NRF_LIB_PATH=$PWD/nerfies_amd/_lib/variants/libnerfies_amd_$v.so; fi
  python bench.py --mode train_bf16 --no-cpu-baseline --burn-in-s 0.5 --steps 40 > gpurun_out/bf16_attr.json 2>/dev/null
  python - $v <<'PY'
import json, sys
d = json.load(open('gpurun_out/bf16_attr.json')); k = d['kernels']
print(f"{sys.argv[1]:8s}: {d['value']/1e3:6.1f} k rays/s  fwd {k['mlp_fwd_coarse']['ms']:.3f}+{k['mlp_fwd_fine']['ms']:.3f} dgrad {k['mlp_dgrad_coarse']['ms']:.3f}+{k['mlp_dgrad_fine']['ms']:.3f} wgrad {k['wgrad_bf16']['ms']:.3f}")
PY
done
unset NRF_LIB_PATH
for c in "12 16" "16 16" "24 16" "16 32"; do
  set -- $c
  NRF_BCOST_CHUNK=$1 NRF_BCOST_SEG=$2 python bench.py --mode train_bf16 --no-cpu-baseline --burn-in-s 0.5 --steps 40 > gpurun_out/bf16_cost.json 2>/dev/null
  python - "$1 $2" <<'PY'
import json, sys
d = json.load(open('gpurun_out/bf16_cost.json')); k = d['kernels']
print(f"chunk/seg {sys.argv[1]:8s}: {d['value']/1e3:6.1f} k rays/s  wgrad_bf16 {k['wgrad_bf16']['ms']:.3f} ms")
PY
done
