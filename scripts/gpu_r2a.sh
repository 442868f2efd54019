set -x
mkdir -p gpurun_out/r2a
nproc > gpurun_out/r2a/nproc.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r2a/pytest_gpu.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r2a/bench_default.log 2>&1
timeout 300 python scripts/exp_miopen_convs.py > gpurun_out/r2a/miopen_convs.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2a/prof_restore -o restore -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2a/rocprof_restore.log 2>&1
