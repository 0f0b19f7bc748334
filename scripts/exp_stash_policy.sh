#!/bin/bash
# A/B of the stash-store / wgrad-load cache policies (variants built by scripts/build_variant.py):
#   base (aux 0) | nt: stores nt | wt: stores sc0 sc1 (write-through) | ntwt: stores sc0 nt sc1 + wgrad loads nt | ntw | w
mkdir -p gpurun_out
for v in base "$@"; do
  if [ $v = base ]; then unset NRF_LIB_PATH; else export NRF_LIB_PATH=$PWD/nerfies_amd/_lib/variants/libnerfies_amd_$v.so; fi
  python bench.py --steps 40 --warmup 5 --burn-in-s 1 --no-cpu-baseline > gpurun_out/exp_policy_$v.json 2> gpurun_out/exp_policy_$v.err
  python - $v <<'PY'
import json, sys
v = sys.argv[1]
try:
  d = json.load(open(f'gpurun_out/exp_policy_{v}.json'))
  k = d['kernels']
  print(f"{v:6s} {d['value']/1e3:7.1f} k rays/s  step {d['ms_per_step']:.3f} ms | " + ' '.join(f"{n} {k[n]['ms']:.3f}" for n in ('mlp_fwd_coarse', 'mlp_fwd_fine', 'mlp_dgrad_coarse', 'mlp_dgrad_fine', 'wgrad')))
except Exception as e:
  print(v, 'FAILED', e, open(f'gpurun_out/exp_policy_{v}.err').read()[-500:])
PY
done
