mkdir -p gpurun_out/r2k
cd /tmp && export TMPDIR=/tmp
export C2M_CONV_TPW=8
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2k/pmc1 -o pmc -- python $GRAFT_REPO_ROOT/scripts/bench_conv.py --only "body 64->64 @640" --iters 6 > $GRAFT_REPO_ROOT/gpurun_out/r2k/pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2k/pmc2 -o pmc -- python $GRAFT_REPO_ROOT/scripts/bench_conv.py --only "body 64->64 @640" --iters 6 > $GRAFT_REPO_ROOT/gpurun_out/r2k/pmc2.log 2>&1
