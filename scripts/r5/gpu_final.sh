#!/bin/bash
# round 5, final GPU call: the whole -m gpu suite on the final sources, the same-box A/B of NRF_OPT_BF16_WGRAD_MERGE (+ its FETCH_SIZE
# pass), then every quoted figure of the round in one go (scripts/gpu_profile_round.sh r05a).
O=gpurun_out; mkdir -p $O/r5f
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -x -q -m gpu -s > $O/r5f/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r5f/pytest_gpu.log
grep -h "one-hop\|\[2 ranks\|bf16 convergence\|bf16 training\|graphed step" $O/r5f/pytest_gpu.log > $O/r5f/parity_report.txt
# ---- A/B: bf16 wgrad groups merged (operands streamed once) vs one group per matrix, same box, back to back, twice ----
for rep in 1 2; do for M in "" "--bf16-wgrad-merge"; do
  tag=$([ -z "$M" ] && echo off || echo on)
  timeout 200 python bench.py --mode train_bf16 $M --steps 50 --warmup 5 --burn-in-s 2 --no-cpu-baseline > $O/r5f/ab_train_bf16_${tag}_$rep.json 2>/dev/null
  timeout 200 python bench.py --mode fullhd --bf16 $M --steps 50 --warmup 5 --burn-in-s 2 --no-cpu-baseline > $O/r5f/ab_fullhd_bf16_${tag}_$rep.json 2>/dev/null
done; done
rm -rf $O/r5f/pmc2
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/r5f/pmc2 -o pmc -- python bench.py --mode train_bf16 --bf16-wgrad-merge --steps 3 --warmup 1 --burn-in-s 0 --no-cpu-baseline > $O/r5f/pmc2.log 2>&1
f=$(find $O/r5f/pmc2 -name '*.db' | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f $O/r5f/train_bf16_merged_pmc_fetch.md; rm -rf $O/r5f/pmc2
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5f/ab_*.json')):
  try:
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); k=d['kernels']
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), 'wgrad_bf16', round(k['wgrad_bf16']['ms'],4))
  except Exception as e: print(f,'ERR',e)
P
bash scripts/gpu_profile_round.sh r05a > $O/r5f/profile_round.log 2>&1
tail -5 $O/r5f/profile_round.log
