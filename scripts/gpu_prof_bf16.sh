#!/bin/bash
# kernel trace + one SQ PMC pass + FETCH/WRITE of the bf16 training line -> gpurun_out/<tag>_train_bf16_*.md
TAG=${1:-x}
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
summ() { f=$(find $1 -name '*.db' | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f $2; }
A="--mode train_bf16 --steps 5 --warmup 1 --burn-in-s 0 --no-cpu-baseline"
rm -rf $O/p1_$TAG $O/p2_$TAG $O/p3_$TAG $O/p4_$TAG
rocprofv3 --kernel-trace --stats -d $O/p1_$TAG -o kt -- python bench.py $A > $O/p1_$TAG.log 2>&1; summ $O/p1_$TAG $O/${TAG}_train_bf16_kernel_stats.md
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/p2_$TAG -o pmc -- python bench.py $A > $O/p2_$TAG.log 2>&1; summ $O/p2_$TAG $O/${TAG}_train_bf16_pmc_sq.md
rocprofv3 --pmc FETCH_SIZE -d $O/p3_$TAG -o pmc -- python bench.py $A > $O/p3_$TAG.log 2>&1; summ $O/p3_$TAG $O/${TAG}_train_bf16_pmc_fetch.md
rocprofv3 --pmc WRITE_SIZE -d $O/p4_$TAG -o pmc -- python bench.py $A > $O/p4_$TAG.log 2>&1; summ $O/p4_$TAG $O/${TAG}_train_bf16_pmc_write.md
grep -h "bf16\|wgrad" $O/${TAG}_train_bf16_*.md | cut -c1-220
