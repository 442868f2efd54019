set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=/root/repo
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 1200 python -m pytest tests -m gpu -q -rA 2>&1 | tail -80 > gpurun_out/pytest_gpu.log
cd /tmp
timeout 300 rocprofv3 -L > $R/gpurun_out/counters.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_corr -o corr -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/rocprof_corr.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $R/gpurun_out/pmc_fetch -o corr -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $R/gpurun_out/pmc_write -o corr -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -f csv -d $R/gpurun_out/pmc_mfma -o corr -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_mfma.log 2>&1
cd $R
find gpurun_out -name "*.csv" | head -30
for f in $(find gpurun_out -name "*.db"); do rm -f $f; done
