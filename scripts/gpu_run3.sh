set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=/root/repo
timeout 1500 python -m pytest tests -m gpu -q -rA 2>&1 | tail -80 > gpurun_out/pytest_gpu.log
timeout 900 python scripts/bench_dcn.py > gpurun_out/bench_dcn.log 2>&1; echo "rc=$?" >> gpurun_out/bench_dcn.log
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_dcn -o dcn -- python $R/scripts/bench_dcn.py --iters 2 > $R/gpurun_out/rocprof_dcn.log 2>&1
cd $R
for f in $(find gpurun_out -name "*.db"); do rm -f $f; done
