set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=/root/repo
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -q -rA 2>&1 | tail -50 > gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1; echo "rc=$?" >> gpurun_out/bench_default.log
timeout 900 python scripts/bench_dcn.py > gpurun_out/bench_dcn.log 2>&1
timeout 900 python scripts/bench_restore.py > gpurun_out/bench_restore.log 2>&1
timeout 600 python scripts/bench_train.py > gpurun_out/bench_train.log 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_corr -o corr -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/rocprof_corr.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_dcn -o dcn -- python $R/scripts/bench_dcn.py --iters 2 > $R/gpurun_out/rocprof_dcn.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $R/gpurun_out/pmc_fetch -o corr -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $R/gpurun_out/pmc_write -o corr -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace -f csv -d $R/gpurun_out/pmc_mfma -o corr -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_mfma.log 2>&1
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace -f csv -d $R/gpurun_out/pmc_dcn -o dcn -- python $R/scripts/bench_dcn.py --iters 1 > $R/gpurun_out/pmc_dcn.log 2>&1
cd $R
for f in $(find gpurun_out -name "*.db"); do rm -f $f; done
