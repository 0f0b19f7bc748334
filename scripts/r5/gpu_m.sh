#!/bin/bash
# round 5, GPU call M: held-out PSNR of fp32 / bf16 / bf16 'mlp' against the step count at the vrig preset's posenc widths.
O=gpurun_out/r5m; mkdir -p $O
timeout 900 python scripts/r5/bf16_warp_gap.py > $O/bf16_warp_gap.json 2> $O/bf16_warp_gap.err; echo rc=$?; grep -v amdgpu $O/bf16_warp_gap.err | tail -8
