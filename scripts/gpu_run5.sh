set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=/root/repo
timeout 1500 python -m pytest tests -m gpu -q -rA 2>&1 | tail -60 > gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench.log 2>&1
timeout 900 python scripts/bench_dcn.py > gpurun_out/bench_dcn.log 2>&1; echo "rc=$?" >> gpurun_out/bench_dcn.log
cd /tmp
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -f csv -d $R/gpurun_out/pmc_dcn -o dcn -- python $R/scripts/bench_dcn.py --iters 1 > $R/gpurun_out/pmc_dcn.log 2>&1
cd $R
