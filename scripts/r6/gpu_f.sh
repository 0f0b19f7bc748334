#!/bin/bash
# round 6, GPU call F: what the prologue VJP and the embedding-gradient atomics cost in se3_bwd_bf16_kernel<false> (experiment builds),
# second cost sweep (narrow-shape chunk term).
O=gpurun_out/r6f; mkdir -p $O
export TMPDIR=/tmp
python scripts/r6/ab_variants.py $O/ab_wbx_fullhd.json --mode fullhd --bf16 -- product wbx1 wbx2 wbx3 product 2> $O/ab.err | tee $O/ab_wbx.txt
timeout 1500 python scripts/r6/cost_sweep.py $O/cost_sweep.json 2> $O/cost_sweep.err | tee $O/cost_sweep.txt
