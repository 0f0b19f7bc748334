mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -q 2>&1 | tail -25
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_quick.json'))
print('rays/s %.0f  ms/step %.3f  step TF %.1f (%.1f%%)' % (d['value'], d['ms_per_step'], d['step_tflops'], 100*d['step_frac_of_fp32_mfma_peak']))
for k,v in d['kernels'].items(): print('  %-18s %8.3f ms x%.0f  %s' % (k, v['ms'], v['launches_per_step'], ('%.1f TF' % v['tflops']) if v['tflops'] else ''))
PY
tail -3 gpurun_out/bench_quick.err
