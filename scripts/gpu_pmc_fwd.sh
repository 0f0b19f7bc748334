export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/pmcf1 gpurun_out/pmcf2
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/pmcf1 -o pmc -- python scripts/exp_fwd.py > gpurun_out/pmcf1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU -d gpurun_out/pmcf2 -o pmc -- python scripts/exp_fwd.py > gpurun_out/pmcf2.log 2>&1
for d in pmcf1 pmcf2; do f=$(find gpurun_out/$d -name '*.db' | head -1); python scripts/rocpd_summary.py $f gpurun_out/$d.md; grep "nerf_mlp_fwd" gpurun_out/$d.md; done
