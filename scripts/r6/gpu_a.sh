#!/bin/bash
# round 6, GPU call A: HBM store / load ceiling in the stash's access pattern; the whole -m gpu suite on the new elastic kernel +
# bench line; the default bench line (certificates); elastic timing in the fullhd / vrig bf16 steps; instruction-cache counters.
O=gpurun_out/r6a; mkdir -p $O
export TMPDIR=/tmp
scripts/micro/_bin/hbm_store_bw > $O/hbm_store_bw.txt 2>&1; cat $O/hbm_store_bw.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json; tail -3 $O/bench_default.err
for m in "fullhd --bf16" "vrig --bf16" "vrig"; do
  n=$(echo $m | tr -d ' -')
  timeout 300 python bench.py --mode $m --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_$n.json 2> $O/bench_$n.err
  python scripts/show_bench.py $O/bench_$n.json | head -24
done
ICC="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU"
timeout 300 rocprofv3 --pmc $ICC -d $O/pmc_ic -o pmc -- python bench.py --mode fullhd --bf16 --steps 3 --warmup 1 --burn-in-s 0 --no-cpu-baseline > $O/pmc_ic.log 2>&1
f=$(find $O/pmc_ic -name '*.db' | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f $O/fullhd_bf16_pmc_icache.md; rm -rf $O/pmc_ic
grep -i "elastic\|se3" $O/fullhd_bf16_pmc_icache.md | head -40
