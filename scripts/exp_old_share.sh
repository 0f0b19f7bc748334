#!/bin/bash
# needs an experiment build: python scripts/build_variant.py exp -DNRF_EXPERIMENT && export NRF_LIB_PATH=nerfies_amd/_lib/variants/libnerfies_amd_exp.so
# uneven static tile split of the fp32 chain kernels: share of a CU's tiles given to its older workgroup
for s in 0 0.5 0.55 0.58 0.62 0.67; do
  NRF_OLD_SHARE=$s python bench.py --no-cpu-baseline --burn-in-s 0.5 --steps 40 > gpurun_out/old_share.json 2>/dev/null
  python - $s <<'PY'
import json, sys
d = json.load(open('gpurun_out/old_share.json')); k = d['kernels']
print(f"share {sys.argv[1]:5s}: {d['value']/1e3:6.1f} k rays/s  fwd {k['mlp_fwd_coarse']['ms']:.3f}+{k['mlp_fwd_fine']['ms']:.3f} dgrad {k['mlp_dgrad_coarse']['ms']:.3f}+{k['mlp_dgrad_fine']['ms']:.3f} wgrad {k['wgrad']['ms']:.3f}")
PY
done
