mkdir -p gpurun_out/r2h
timeout 600 python -m pytest tests/test_conv_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/r2h/pytest_conv.log
timeout 300 python scripts/bench_conv.py > gpurun_out/r2h/bench_conv_all.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2h/pmc1 -o pmc -- python $GRAFT_REPO_ROOT/scripts/bench_conv.py --only "body 64->64 @640" --iters 3 > $GRAFT_REPO_ROOT/gpurun_out/r2h/pmc1.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_restoration_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/r2h/pytest_rest.log
