mkdir -p gpurun_out/r2g
for st in 0 1 2 4 8; do
  export C2M_CONV_STAGGER=$st
  echo "== stagger $st" >> gpurun_out/r2g/bench_conv.log
  timeout 120 python scripts/bench_conv.py --only "body" >> gpurun_out/r2g/bench_conv.log 2>&1
done
unset C2M_CONV_STAGGER
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2g/pmc1 -o pmc -- python $GRAFT_REPO_ROOT/scripts/bench_conv.py --only "body 64->64 @640" --iters 3 > $GRAFT_REPO_ROOT/gpurun_out/r2g/pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2g/pmc2 -o pmc -- python $GRAFT_REPO_ROOT/scripts/bench_conv.py --only "body 64->64 @640" --iters 3 > $GRAFT_REPO_ROOT/gpurun_out/r2g/pmc2.log 2>&1
