#!/bin/bash
# round 4, call F: the whole GPU suite as the driver runs it + smoke()
O=gpurun_out/r4f; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -q -m gpu --durations=15 > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"
grep -h "^FAILED\|^ERROR\|^E  " $O/tests.log | head -40
grep -A18 "slowest" $O/tests.log | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
