#!/bin/bash
# NOTE: the f32nostash variant needs an early `return;` at the top of buf_store4 (chain_common.h) under -DNRF_EXP_NOSTASH; the switch is
# not kept in the tree (it would change the kernel-source hash the HBM traffic table is stamped with).
for v in base f32nostash; do
  if [ $v = base ]; then unset NRF_LIB_PATH; else export NRF_LIB_PATH=$PWD/nerfies_amd/_lib/variants/libnerfies_amd_$v.so; fi
  python bench.py --no-cpu-baseline --burn-in-s 0.5 --steps 40 > gpurun_out/f32_attr.json 2>/dev/null
  python - $v <<'PY'
import json, sys
d = json.load(open('gpurun_out/f32_attr.json')); k = d['kernels']
print(f"{sys.argv[1]:10s}: {d['value']/1e3:6.1f} k rays/s  fwd {k['mlp_fwd_coarse']['ms']:.3f}+{k['mlp_fwd_fine']['ms']:.3f} dgrad {k['mlp_dgrad_coarse']['ms']:.3f}+{k['mlp_dgrad_fine']['ms']:.3f} wgrad {k['wgrad']['ms']:.3f}")
PY
done
