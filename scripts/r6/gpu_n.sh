#!/bin/bash
# round 6, GPU call N: the split-bf16 (bf16x3) inference chains -- parity tests, eval lines next to the float32 ones, kernel trace
O=gpurun_out/r6n; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_bf16x3.py -m gpu -q -s -p no:cacheprovider > $O/x3_tests.txt 2>&1; tail -3 $O/x3_tests.txt
for m in "--split-bf16" "--split-bf16 --warp --frame" "" "--warp --frame"; do
  n=$(echo "eval$m" | tr -d ' -')
  timeout 300 python bench.py --mode eval $m --steps 30 --warmup 3 > $O/bench_$n.json 2> $O/bench_$n.err
  python scripts/show_bench.py $O/bench_$n.json | head -12
done
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_x3 -o kt -- python bench.py --mode eval --split-bf16 --steps 10 --warmup 2 --burn-in-s 0 > $O/prof_x3.log 2>&1
f=$(find $O/prof_x3 -name '*.db' | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f $O/eval_x3_kernel_stats.md; rm -rf $O/prof_x3
head -12 $O/eval_x3_kernel_stats.md
SQ="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY"
timeout 300 rocprofv3 --pmc $SQ -d $O/pmc_x3 -o pmc -- python bench.py --mode eval --split-bf16 --steps 3 --warmup 1 --burn-in-s 0 > $O/pmc_x3.log 2>&1
f=$(find $O/pmc_x3 -name '*.db' | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f $O/eval_x3_pmc_sq.md; rm -rf $O/pmc_x3
grep "x3" $O/eval_x3_pmc_sq.md | head -20
