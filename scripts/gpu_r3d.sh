# round 3, call F: PMC counters of the split kernel on the 64->64 @640 body layer
set -x
mkdir -p gpurun_out/r3f
O=$GRAFT_REPO_ROOT/gpurun_out/r3f
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
CMD="python $R/scripts/bench_conv.py --algo split --iters 3 --only 64->64@640x"
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VMEM" \
            "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  n=$(echo $pass | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $pass --kernel-trace -f csv -d $O/pmc_$n -o x -- python $R/scripts/bench_conv.py --algo split --iters 3 --only "body 64->64 @640" > $O/pmc_$n.log 2>&1
done
cd $R
python scripts/pmc_kernel.py gpurun_out/r3f conv3x3_split > gpurun_out/r3f/summary.txt 2>&1
cat gpurun_out/r3f/summary.txt
find gpurun_out/r3f -name "*.db" -delete
