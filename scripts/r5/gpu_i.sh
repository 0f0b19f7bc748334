#!/bin/bash
# round 5, GPU call I: the tests edited after the last full pass (warp-on bf16 convergence at the vrig preset's posenc widths; the
# bench self-launch with the hardened strong-scaling section) and the SQ counters of the 128-ray step (half-tile forward kernels).
O=gpurun_out/r5i; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_bf16_convergence.py tests/test_gpu_rccl.py -q -m gpu -s > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
grep -h "bf16 training\|bf16 convergence" $O/pytest.log | cut -c1-400
SQ="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
rm -rf $O/pmc1
timeout 300 rocprofv3 --pmc $SQ -d $O/pmc1 -o pmc -- python bench.py --rays-per-gpu 128 --steps 3 --warmup 1 --burn-in-s 0 --no-cpu-baseline > $O/pmc1.log 2>&1
f=$(find $O/pmc1 -name '*.db' | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f $O/r05_train128_pmc_sq.md; rm -rf $O/pmc1
grep -i "fwd32\|nerf_mlp_bwd\|wgrad_kernel" $O/r05_train128_pmc_sq.md | head -30 | cut -c1-200
