#!/bin/bash
# round 5, GPU call E: cost of the merged bf16 wgrad shapes in the stream-K model (NRF_BCOST_MERGED, experiment build), then the tests
# touched since call D.
O=gpurun_out/r5e; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
for X in 0 6 10 16 24 32; do
  NRF_LIB_PATH=nerfies_amd/_lib/variants/libnerfies_amd_exp.so NRF_BCOST_MERGED=$X timeout 200 python bench.py --mode train_bf16 --steps 40 --warmup 5 --burn-in-s 1 --no-cpu-baseline > $O/sweep_$X.json 2> $O/sweep_$X.err
done
timeout 200 python bench.py --mode fullhd --bf16 --steps 40 --warmup 5 --burn-in-s 1 --no-cpu-baseline > $O/fullhd_bf16.json 2> $O/fullhd_bf16.err
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5e/*.json')):
  try:
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); k=d['kernels']
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), 'wgrad_bf16', round(k['wgrad_bf16']['ms'],4), 'reduce', round(k['grad_reduce']['ms'],4))
  except Exception as e: print(f,'ERR',e)
P
timeout 900 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_graph_step.py tests/test_gpu_bf16_train.py tests/test_gpu_contract.py tests/test_gpu_chain32.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
