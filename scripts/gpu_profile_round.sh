#!/bin/bash
# One command regenerates every figure DESIGN.md / README.md / BASELINE.md quote for a round:
#   scripts/gpu_profile_round.sh r04a ["modes"]     (on the GPU box, e.g. through gpurun)
# -> gpurun_out/<tag>_*.json|md ; scripts/collect_profiles.py <tag> r04 copies them into profiles/ (and merges the per-workload
# HBM tables into profiles/hbm_traffic.json).
# Then scripts/stamp_traffic.py r04 fills `roofline.traffic` of the copied bench lines from the PMC passes of the SAME run (the
# bench line of a workload is written before its PMC passes exist) and scripts/fill_round4_docs.py writes profiles/r04_summary.md.
# Passes are separate processes: bench line, rocprofv3 kernel trace, three PMC passes (SQ / FETCH_SIZE / WRITE_SIZE never
# share a pass, never combined with a trace domain).
TAG=${1:-r04}
MODES=${2:-"train train_bf16 vrig vrig_bf16 fullhd fullhd_bf16 eval eval_bf16 eval_x3 eval_warp eval_warp_bf16 eval_warp_x3 eval_warp_x3mlp train128 train128_graph sustained sustained_bf16"}
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
summ() { f=$(find $1 -name '*.db' | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f $2; }
# gpurun copies back at most 64 MiB: the rocpd databases are dropped once summarised (hbm traffic is extracted first)
clean() { rm -rf "$@"; }
SQ="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
# LDS side of the bf16 chain kernels (its own pass): instructions, array-busy cycles, conflict cycles
LDSC="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA"
ICC="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CYCLES SQ_WAVE_CYCLES"
for mode in $MODES; do
  PMC=1; TRACE=1; STEPS=50
  case $mode in
    train)          ARGS="";                                   SUF="" ;;
    train_bf16)     ARGS="--mode train_bf16";                  SUF="_train_bf16" ;;
    vrig)           ARGS="--mode vrig";                        SUF="_vrig" ;;
    vrig_bf16)      ARGS="--mode vrig --bf16";                 SUF="_vrig_bf16" ;;
    fullhd)         ARGS="--mode fullhd";                      SUF="_fullhd" ;;
    fullhd_bf16)    ARGS="--mode fullhd --bf16";               SUF="_fullhd_bf16" ;;
    eval)           ARGS="--mode eval";                        SUF="_eval"; PMC=0 ;;
    eval_bf16)      ARGS="--mode eval --bf16";                 SUF="_eval_bf16"; PMC=0 ;;
    eval_x3)        ARGS="--mode eval --split-bf16";           SUF="_eval_x3"; PMC=0 ;;
    eval_warp_x3)   ARGS="--mode eval --warp --frame --split-bf16"; SUF="_eval_warp_x3"; PMC=0 ;;
    eval_warp_x3mlp) ARGS="--mode eval --warp --split-bf16 --warp-f32"; SUF="_eval_warp_x3mlp"; PMC=0; TRACE=0 ;;
    eval_warp)      ARGS="--mode eval --warp --frame";         SUF="_eval_warp"; PMC=0 ;;
    eval_warp_bf16) ARGS="--mode eval --warp --frame --bf16";  SUF="_eval_warp_bf16"; PMC=0 ;;
    train128)       ARGS="--rays-per-gpu 128";                 SUF="_train128"; PMC=0; TRACE=0 ;;
    train128_graph) ARGS="--rays-per-gpu 128 --graph";         SUF="_train128_graph"; PMC=0; TRACE=0 ;;
    # 20+ s timed windows: the default sub-second window is steady state (fp32), and where the board's power limit puts the bf16 step
    sustained)      ARGS="";                                   SUF="_sustained"; PMC=0; TRACE=0; STEPS=3000 ;;
    sustained_bf16) ARGS="--mode train_bf16";                  SUF="_sustained_bf16"; PMC=0; TRACE=0; STEPS=16000 ;;
  esac
  # LIGHT="mode ..." (environment): bench line only for those workloads (when the GPU budget does not cover every pass)
  case " $LIGHT " in *" $mode "*) PMC=0; TRACE=0 ;; esac
  NOCPU="--no-cpu-baseline"; [ "$mode" = train ] && NOCPU=""
  timeout 900 python bench.py $ARGS --steps $STEPS --warmup 5 $NOCPU > $O/${TAG}_bench${SUF}.json 2> $O/${TAG}_bench${SUF}.err
  head -c 300 $O/${TAG}_bench${SUF}.json; echo
  [ $TRACE = 0 ] && continue
  rm -rf $O/prof_${TAG}${SUF} $O/pmc1_${TAG}${SUF} $O/pmc2_${TAG}${SUF} $O/pmc3_${TAG}${SUF}
  PARGS=${ARGS/ --frame/}   # the trace / counter passes time chunks only: a whole frame under PMC serialisation took > 30 min once (round 4)
  # ... and the headline workload ALONE: since round 6 the default line also runs the secondary workloads, whose launches of the same
  # kernel symbols (other shapes) would be averaged into the per-launch times and bytes of the config-A kernels
  [ "$mode" = train ] && PARGS="$PARGS --no-extras"
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_${TAG}${SUF} -o kt -- python bench.py $PARGS --steps 10 --warmup 2 --burn-in-s 0 --no-cpu-baseline > $O/prof_${TAG}${SUF}.log 2>&1
  summ $O/prof_${TAG}${SUF} $O/${TAG}${SUF}_kernel_stats.md
  timeout 300 rocprofv3 --pmc $SQ -d $O/pmc1_${TAG}${SUF} -o pmc -- python bench.py $PARGS --steps 3 --warmup 1 --burn-in-s 0 --no-cpu-baseline > $O/pmc1_${TAG}${SUF}.log 2>&1
  summ $O/pmc1_${TAG}${SUF} $O/${TAG}${SUF}_pmc_sq.md
  case $mode in train|train_bf16|eval_bf16)   # instruction fetch: the unrolled bf16 chains are ~100 KiB of code (cold start of every launch)
    timeout 300 rocprofv3 --pmc $ICC -d $O/pmc5_${TAG}${SUF} -o pmc -- python bench.py $PARGS --steps 3 --warmup 1 --burn-in-s 0 --no-cpu-baseline > $O/pmc5_${TAG}${SUF}.log 2>&1
    summ $O/pmc5_${TAG}${SUF} $O/${TAG}${SUF}_pmc_icache.md; clean $O/pmc5_${TAG}${SUF} ;;
  esac
  case $mode in *bf16*|*x3*)
    timeout 300 rocprofv3 --pmc $LDSC -d $O/pmc4_${TAG}${SUF} -o pmc -- python bench.py $PARGS --steps 3 --warmup 1 --burn-in-s 0 --no-cpu-baseline > $O/pmc4_${TAG}${SUF}.log 2>&1
    summ $O/pmc4_${TAG}${SUF} $O/${TAG}${SUF}_pmc_lds.md; clean $O/pmc4_${TAG}${SUF} ;;
  esac
  if [ $PMC = 1 ]; then
    timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc2_${TAG}${SUF} -o pmc -- python bench.py $PARGS --steps 3 --warmup 1 --burn-in-s 0 --no-cpu-baseline > $O/pmc2_${TAG}${SUF}.log 2>&1
    timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc3_${TAG}${SUF} -o pmc -- python bench.py $PARGS --steps 3 --warmup 1 --burn-in-s 0 --no-cpu-baseline > $O/pmc3_${TAG}${SUF}.log 2>&1
    summ $O/pmc2_${TAG}${SUF} $O/${TAG}${SUF}_pmc_fetch.md
    summ $O/pmc3_${TAG}${SUF} $O/${TAG}${SUF}_pmc_write.md
    fdb=$(find $O/pmc2_${TAG}${SUF} -name '*.db' | head -1); wdb=$(find $O/pmc3_${TAG}${SUF} -name '*.db' | head -1)
    [ -n "$fdb" ] && [ -n "$wdb" ] && python scripts/make_hbm_traffic.py $fdb $wdb $O/${TAG}${SUF}_hbm_traffic.json \
      "profiles/${TAG}${SUF}_pmc_fetch.md, profiles/${TAG}${SUF}_pmc_write.md (bench.py $ARGS --steps 3)"
  fi
  clean $O/prof_${TAG}${SUF} $O/pmc1_${TAG}${SUF} $O/pmc2_${TAG}${SUF} $O/pmc3_${TAG}${SUF}
done
ls $O/${TAG}_*
