mkdir -p gpurun_out
NRF_REGCHAIN=1 timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "forward or render or graphed or invariants" 2>&1 | tail -12
for cfg in "NRF_X=0" "NRF_REGCHAIN=1"; do
  env $cfg python bench.py --mode eval --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/b.json 2> gpurun_out/b.err
  python - "$cfg" <<'PY'
import json, sys
d=json.load(open('gpurun_out/b.json'))
k=d['kernels']
print('%-20s %.0f rays/s  %.3f ms | ' % (sys.argv[1], d['value'], d['ms_per_step']) + ' '.join('%s %.3f (%s)' % (n, v['ms'], ('%.0f TF' % v['tflops']) if v['tflops'] else '-') for n, v in k.items() if 'mlp' in n))
PY
done
tail -3 gpurun_out/b.err
