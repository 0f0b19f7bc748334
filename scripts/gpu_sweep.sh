mkdir -p gpurun_out
run() { echo "== $*"; env "$@" python bench.py --steps 15 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('rays/s %.0f ms %.3f wgrad %.3f ms' % (d['value'], d['ms_per_step'], k['wgrad']['ms']))"; }
run NRF_COST_VEC256=0.10
run NRF_COST_VEC256=0.20 NRF_COST_VEC128=0.12
run NRF_COST_VEC256=0.35 NRF_COST_VEC128=0.2
run NRF_COST_VEC256=0.5 NRF_COST_VEC128=0.3
run NRF_COST_VEC256=0.2 NRF_COST_VEC128=0.12 NRF_COST_PE=0.6
run NRF_COST_VEC256=0.2 NRF_COST_VEC128=0.12 NRF_COST_PE=0.3
run NRF_COST_VEC256=0.2 NRF_COST_VEC128=0.12 NRF_COST_RGBH=0.8
run NRF_COST_VEC256=0.2 NRF_COST_VEC128=0.12 NRF_COST_RGBH=0.5
