# rocprofv3 kernel-trace stats of the two secondary workloads (eval forward, gpu_vrig_paper training shape)
TAG=${1:-r01_d}
mkdir -p gpurun_out
export TMPDIR=/tmp
for mode in eval vrig; do
  rm -rf gpurun_out/prof_${TAG}_${mode}
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${TAG}_${mode} -o kt -- python bench.py --mode ${mode} --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/prof_${TAG}_${mode}.log 2>&1
  f=$(find gpurun_out/prof_${TAG}_${mode} -name '*.db' | head -1)
  [ -n "$f" ] && python scripts/rocpd_summary.py $f gpurun_out/${TAG}_${mode}_prof.md
done
ls gpurun_out/${TAG}_*_prof.md
