# Round-end validation + measurement on one MI355X: everything DESIGN.md / profiles/ quote comes from this script.
set -x
mkdir -p gpurun_out/final
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 1800 python -m pytest tests -m gpu -q -rA 2>&1 | tail -90 > $O/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.log 2>&1; echo "rc=$?" >> $O/bench_default.log
timeout 300 python bench.py --workload corr --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_corr.log 2>&1
timeout 300 python scripts/bench_conv.py > $O/bench_conv.log 2>&1
echo "--- with fast=True (the decoder's setting: Winograd F(4,3) where the map is a multiple of 64 pixels wide)" >> $O/bench_conv.log
timeout 300 python scripts/bench_conv.py --fast --only "64 @" >> $O/bench_conv.log 2>&1
timeout 600 python scripts/bench_dcn.py > $O/bench_dcn.log 2>&1
timeout 600 python scripts/bench_train.py > $O/bench_train.log 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_step -o step -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/rocprof_step.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace -f csv -d $O/pmc_mfma -o step -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $O/pmc_mfma.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $O/pmc_fetch -o step -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $O/pmc_write -o step -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $O/pmc_write.log 2>&1
cd $R
for f in $(find gpurun_out/final -name "*.db"); do rm -f $f; done
# the per-dispatch traces are large: keep only the stats / counter CSVs; then: python scripts/summarize_step.py r02_final 4
find gpurun_out/final -name "*kernel_trace.csv" -size +8M -delete
