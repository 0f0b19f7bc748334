mkdir -p gpurun_out
set -x
python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -3
python bench.py --steps 30 --warmup 5 > gpurun_out/bench_r1_b.json 2> gpurun_out/bench_r1_b.err; cat gpurun_out/bench_r1_b.json | head -c 3000
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b -o kt -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof_b_stdout.txt 2>&1
ls -R gpurun_out/prof_b | head -30
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/pmc_b -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_b_stdout.txt 2>&1
ls -R gpurun_out/pmc_b | head
