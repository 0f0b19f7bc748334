#!/bin/bash
# One command regenerates every figure DESIGN.md / README.md / BASELINE.md quote for a round:
#   scripts/gpu_profile_round.sh r02     (on the GPU box, e.g. through gpurun)
# -> gpurun_out/<tag>_*.json|md ; copy the ones to be judged into profiles/ and commit them.
# Passes are separate processes: bench line, rocprofv3 kernel trace, three PMC passes (SQ / FETCH_SIZE / WRITE_SIZE never
# share a pass, never combined with a trace domain), then the secondary workloads.
TAG=${1:-r02}
MODES=${2:-"train eval eval_bf16 vrig vrig_bf16 train_bf16"}
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
summ() { f=$(find $1 -name '*.db' | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f $2; }
# gpurun copies back at most 64 MiB: the rocpd databases are dropped once summarised (hbm traffic is extracted first)
clean() { rm -rf "$@"; }
SQ="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
for mode in $MODES; do
  case $mode in
    train)      ARGS="";                         ENV="";               SUF="" ;;
    eval)       ARGS="--mode eval";              ENV="";               SUF="_eval" ;;
    eval_bf16)  ARGS="--mode eval";              ENV="BENCH_BF16=1";   SUF="_eval_bf16" ;;
    vrig)       ARGS="--mode vrig";              ENV="";               SUF="_vrig" ;;
    vrig_bf16)  ARGS="--mode vrig";              ENV="BENCH_BF16=1";   SUF="_vrig_bf16" ;;
    train_bf16) ARGS="--mode train_bf16";        ENV="";               SUF="_train_bf16" ;;
  esac
  NOCPU="--no-cpu-baseline"; [ "$mode" = train ] && NOCPU=""
  env $ENV python bench.py $ARGS --steps 50 --warmup 5 $NOCPU > $O/${TAG}_bench${SUF}.json 2> $O/${TAG}_bench${SUF}.err
  head -c 400 $O/${TAG}_bench${SUF}.json; echo
  rm -rf $O/prof_${TAG}${SUF} $O/pmc1_${TAG}${SUF} $O/pmc2_${TAG}${SUF} $O/pmc3_${TAG}${SUF}
  env $ENV rocprofv3 --kernel-trace --stats -d $O/prof_${TAG}${SUF} -o kt -- python bench.py $ARGS --steps 10 --warmup 2 --burn-in-s 0 --no-cpu-baseline > $O/prof_${TAG}${SUF}.log 2>&1
  summ $O/prof_${TAG}${SUF} $O/${TAG}${SUF}_kernel_stats.md
  env $ENV rocprofv3 --pmc $SQ -d $O/pmc1_${TAG}${SUF} -o pmc -- python bench.py $ARGS --steps 3 --warmup 1 --burn-in-s 0 --no-cpu-baseline > $O/pmc1_${TAG}${SUF}.log 2>&1
  summ $O/pmc1_${TAG}${SUF} $O/${TAG}${SUF}_pmc_sq.md
  if [ "$mode" = train ] || [ "$mode" = train_bf16 ]; then
    env $ENV rocprofv3 --pmc FETCH_SIZE -d $O/pmc2_${TAG}${SUF} -o pmc -- python bench.py $ARGS --steps 3 --warmup 1 --burn-in-s 0 --no-cpu-baseline > $O/pmc2_${TAG}${SUF}.log 2>&1
    env $ENV rocprofv3 --pmc WRITE_SIZE -d $O/pmc3_${TAG}${SUF} -o pmc -- python bench.py $ARGS --steps 3 --warmup 1 --burn-in-s 0 --no-cpu-baseline > $O/pmc3_${TAG}${SUF}.log 2>&1
    summ $O/pmc2_${TAG}${SUF} $O/${TAG}${SUF}_pmc_fetch.md
    summ $O/pmc3_${TAG}${SUF} $O/${TAG}${SUF}_pmc_write.md
    fdb=$(find $O/pmc2_${TAG}${SUF} -name '*.db' | head -1); wdb=$(find $O/pmc3_${TAG}${SUF} -name '*.db' | head -1)
    [ -n "$fdb" ] && [ -n "$wdb" ] && python scripts/make_hbm_traffic.py $fdb $wdb $O/${TAG}${SUF}_hbm_traffic.json \
      "profiles/${TAG}${SUF}_pmc_fetch.md, profiles/${TAG}${SUF}_pmc_write.md (bench.py $ARGS --steps 3)"
  fi
  clean $O/prof_${TAG}${SUF} $O/pmc1_${TAG}${SUF} $O/pmc2_${TAG}${SUF} $O/pmc3_${TAG}${SUF}
done
ls $O/${TAG}_*
