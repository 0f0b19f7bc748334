mkdir -p gpurun_out
# needs an experiment build: python scripts/build_variant.py exp -DNRF_EXPERIMENT && export NRF_LIB_PATH=nerfies_amd/_lib/variants/libnerfies_amd_exp.so
for cfg in "NRF_GRID_MUL=2" "NRF_GRID_MUL=1"; do
  env $cfg python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/b.json 2> gpurun_out/b.err
  python - "$cfg" <<'PY'
import json, sys
d=json.load(open('gpurun_out/b.json'))
k=d['kernels']
print('%-36s %.0f rays/s  %.3f ms | fwd c %.3f f %.3f | dgrad c %.3f f %.3f | wgrad %.3f' % (sys.argv[1], d['value'], d['ms_per_step'],
  k['mlp_fwd_coarse']['ms'], k['mlp_fwd_fine']['ms'], k['mlp_dgrad_coarse']['ms'], k['mlp_dgrad_fine']['ms'], k['wgrad']['ms']))
PY
done
