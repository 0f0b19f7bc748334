#!/bin/bash
# Runs every GPU test file in its own process (an abort in one does not hide the others); logs under gpurun_out/$1
tag=${1:-gpu}
mkdir -p gpurun_out/$tag
for f in ${FILES:-tests/test_g*.py}; do
  n=$(basename $f .py)
  timeout 900 python -m pytest $f -m gpu -q -v -x --durations=5 > gpurun_out/$tag/$n.log 2>&1
  echo "$n rc=$? $(grep -E '^(=+ )?[0-9]+ (passed|failed)|passed|failed' gpurun_out/$tag/$n.log | tail -1)"
done
