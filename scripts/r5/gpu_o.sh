#!/bin/bash
# round 5, GPU call O: the one-GPU side of the strong-scaling curve in the bf16 training mode (1024-ray batch shared by 8 / 4 / 2 GPUs).
O=gpurun_out/r5o; mkdir -p $O
for R in 128 256 512; do
  timeout 200 python bench.py --mode train_bf16 --rays-per-gpu $R --steps 200 --warmup 10 --burn-in-s 2 --no-cpu-baseline > $O/r05_bench_train_bf16_$R.json 2>/dev/null
  timeout 200 python bench.py --mode train_bf16 --rays-per-gpu $R --steps 200 --warmup 10 --burn-in-s 2 --no-cpu-baseline --graph > $O/r05_bench_train_bf16_${R}_graph.json 2>/dev/null
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5o/*.json')):
  try:
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), round(d['sum_of_kernels_ms'],4))
  except Exception as e: print(f,'ERR',e)
P
