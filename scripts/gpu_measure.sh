# Round measurement: bench line + rocprofv3 kernel stats + PMC passes (separate runs), summaries -> gpurun_out/
TAG=${1:-r01_b}
mkdir -p gpurun_out
export TMPDIR=/tmp
python bench.py --steps 30 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
head -c 600 gpurun_out/${TAG}_bench.json; echo
rm -rf gpurun_out/prof_${TAG} gpurun_out/pmc1_${TAG} gpurun_out/pmc2_${TAG} gpurun_out/pmc3_${TAG}
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${TAG} -o kt -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof_${TAG}.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/pmc1_${TAG} -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc1_${TAG}.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc2_${TAG} -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc2_${TAG}.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc3_${TAG} -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc3_${TAG}.log 2>&1
for d in prof pmc1 pmc2 pmc3; do
  f=$(find gpurun_out/${d}_${TAG} -name '*.db' | head -1)
  [ -n "$f" ] && python scripts/rocpd_summary.py $f gpurun_out/${TAG}_${d}.md
done
ls gpurun_out/${TAG}_*
