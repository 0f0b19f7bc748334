#!/bin/bash
# round 5, GPU call N: one rank's share of a 960 x 540 frame on 8 GPUs under render_image's two tile_parallel modes (launches only).
O=gpurun_out/r5n; mkdir -p $O
timeout 600 python scripts/r5/eval_band_vs_chunk.py > $O/eval_band_vs_chunk.json 2> $O/eval_band_vs_chunk.err; echo rc=$?; grep -v amdgpu $O/eval_band_vs_chunk.err | tail -4
