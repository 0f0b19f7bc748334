#!/bin/bash
# round 4, call B: new parity tests (one-hop reference vectors, configs[0] preset), eval with the warp on + whole-frame timing,
# LDS / issue counters of the rebuilt bf16 chain kernels
O=gpurun_out/r4b; mkdir -p $O
export TMPDIR=/tmp
for f in tests/test_gpu_reference_onehop.py tests/test_gpu_datasets.py tests/test_gpu_rccl.py tests/test_gpu_graph_step.py; do
  n=$(basename $f .py)
  timeout 900 python -m pytest $f -m gpu -q -x -s --durations=3 > $O/$n.log 2>&1
  echo "$n rc=$? $(grep -E 'passed|failed|error' $O/$n.log | tail -1)"
done
grep -h "one-hop" $O/test_gpu_reference_onehop.log
for v in "" "--bf16"; do
  s=$( [ -z "$v" ] && echo f32 || echo bf16 )
  timeout 300 python bench.py --mode eval --warp --frame $v --steps 20 --warmup 3 --burn-in-s 1 > $O/bench_eval_warp_$s.json 2> $O/bench_eval_warp_$s.err; echo "eval_warp_$s rc=$?"
done
timeout 300 python bench.py --mode eval --frame --bf16 --steps 20 --warmup 3 --burn-in-s 1 > $O/bench_eval_frame_bf16.json 2> $O/bench_eval_frame_bf16.err; echo "eval_frame_bf16 rc=$?"
python scripts/show_bench.py $O/bench_eval_warp_f32.json $O/bench_eval_warp_bf16.json $O/bench_eval_frame_bf16.json 2>&1 | tail -50
python - <<'PY'
import json
for s in ('warp_f32','warp_bf16','frame_bf16'):
  try: print(s, json.load(open(f'gpurun_out/r4b/bench_eval_{s}.json'))['frame'])
  except Exception as e: print(s, 'unreadable', e)
PY
summ() { f=$(find $1 -name '*.db' | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f $2; }
LDS="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
ISS="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU"
for m in "eval_bf16:--mode eval --bf16" "train_bf16:--mode train_bf16"; do
  tag=${m%%:*}; args=${m#*:}
  rocprofv3 --pmc $LDS -d $O/pmc_lds_$tag -o pmc -- python bench.py $args --steps 3 --warmup 1 --burn-in-s 0 --no-cpu-baseline > $O/pmc_lds_$tag.log 2>&1
  summ $O/pmc_lds_$tag $O/${tag}_pmc_lds.md
  rocprofv3 --pmc $ISS -d $O/pmc_iss_$tag -o pmc -- python bench.py $args --steps 3 --warmup 1 --burn-in-s 0 --no-cpu-baseline > $O/pmc_iss_$tag.log 2>&1
  summ $O/pmc_iss_$tag $O/${tag}_pmc_issue.md
  rm -rf $O/pmc_lds_$tag $O/pmc_iss_$tag
done
grep -h "bf16_kernel" $O/*_pmc_lds.md $O/*_pmc_issue.md | cut -c1-400
tail -2 $O/*.err | tail -20
