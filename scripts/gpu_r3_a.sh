#!/bin/bash
# round-3 first GPU pass: full GPU test suite + the three training bench lines + wgrad cost-model calibration
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/r3a_pytest.log 2>&1; echo "pytest rc $?" >> $O/r3a_pytest.log
tail -15 $O/r3a_pytest.log
for m in train vrig train_bf16; do
  ARGS="--mode $m"; [ $m = train ] && ARGS=""
  timeout 300 python bench.py $ARGS --steps 30 --warmup 5 --no-cpu-baseline > $O/r3a_bench_$m.json 2> $O/r3a_bench_$m.err
  python scripts/show_bench.py $O/r3a_bench_$m.json
done
BENCH_BF16=1 timeout 300 python bench.py --mode vrig --steps 30 --warmup 5 --no-cpu-baseline > $O/r3a_bench_vrig_bf16.json 2> $O/r3a_bench_vrig_bf16.err
python scripts/show_bench.py $O/r3a_bench_vrig_bf16.json
timeout 200 python scripts/wgrad_calib.py > $O/r3a_calib.txt 2>&1; cat $O/r3a_calib.txt
timeout 200 python scripts/wgrad_calib_vrig.py > $O/r3a_calib_vrig.txt 2>&1; cat $O/r3a_calib_vrig.txt
