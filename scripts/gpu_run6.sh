set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=/root/repo
timeout 900 python scripts/bench_dcn.py > gpurun_out/bench_dcn.log 2>&1; echo "rc=$?" >> gpurun_out/bench_dcn.log
timeout 900 python scripts/bench_restore.py > gpurun_out/bench_restore.log 2>&1; echo "rc=$?" >> gpurun_out/bench_restore.log
cd /tmp
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -f csv -d $R/gpurun_out/pmc_corr2 -o corr -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_corr2.log 2>&1
timeout 900 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS --kernel-trace -f csv -d $R/gpurun_out/pmc_corr3 -o corr -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_corr3.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_restore -o restore -- python $R/scripts/bench_restore.py --steps 2 > $R/gpurun_out/rocprof_restore.log 2>&1
cd $R
for f in $(find gpurun_out -name "*.db"); do rm -f $f; done
