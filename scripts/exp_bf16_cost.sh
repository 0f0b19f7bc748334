#!/bin/bash
# bf16 wgrad partition cost model sweep
for c in "3 24" "0 24" "6 24" "3 8" "3 64" "8 16"; do
  set -- $c
  NRF_BCOST_CHUNK=$1 NRF_BCOST_SEG=$2 python bench.py --mode train_bf16 --no-cpu-baseline --burn-in-s 0.5 --steps 40 > gpurun_out/bf16_cost.json 2>/dev/null
  python - "$1 $2" <<'PY'
import json, sys
d = json.load(open('gpurun_out/bf16_cost.json')); k = d['kernels']
print(f"chunk/seg {sys.argv[1]:8s}: {d['value']/1e3:6.1f} k rays/s  wgrad_bf16 {k['wgrad_bf16']['ms']:.3f} ms  fwd {k['mlp_fwd_fine']['ms']:.3f} dgrad {k['mlp_dgrad_fine']['ms']:.3f}")
PY
done
