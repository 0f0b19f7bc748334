#!/bin/bash
# round 4, final call: the whole GPU suite as the driver runs it, smoke(), then the profile round (scripts/gpu_profile_round.sh r04a)
O=gpurun_out/r4final; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -q -m gpu --durations=8 > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"
grep -h "^FAILED\|^ERROR" $O/tests.log | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash scripts/gpu_profile_round.sh r04a > gpurun_out/r04a_round.log 2>&1; tail -3 gpurun_out/r04a_round.log
