#!/bin/bash
# round 5, last GPU call: the whole -m gpu suite in ONE pass as the driver runs it, the rest of the one-GPU strong-scaling curve
# (256 / 512 rays per GPU), the kernel trace of the 128-ray step (the half-tile forward kernels in rocprofv3's own table), and the
# self-launching N=2 line on the final sources.
O=gpurun_out/r5h; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
for R in 256 512; do
  timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --rays-per-gpu $R > $O/r05_bench_train$R.json 2> $O/bench_train$R.err
done
rm -rf $O/prof128
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof128 -o kt -- python bench.py --rays-per-gpu 128 --steps 10 --warmup 2 --burn-in-s 0 --no-cpu-baseline > $O/prof128.log 2>&1
f=$(find $O/prof128 -name '*.db' | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f $O/r05_train128_kernel_stats.md; rm -rf $O/prof128
( timeout 600 python bench.py --gpus 2 --steps 10 --warmup 2 --burn-in-s 1 > $O/r05_bench_gpus2_one_gpu_lease.json 2> $O/bench_gpus2.err; echo "gpus2 rc=$?" )
head -12 $O/r05_train128_kernel_stats.md | cut -c1-200
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5h/*.json')):
  try:
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), d.get('n_gpus'), d.get('oversubscribed'), d.get('replica_param_checksums_agree'))
  except Exception as e: print(f,'ERR',e)
P
