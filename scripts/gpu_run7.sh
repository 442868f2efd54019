set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=/root/repo
timeout 1500 python -m pytest tests -m gpu -q -rA 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench.log 2>&1
timeout 900 python scripts/bench_restore.py > gpurun_out/bench_restore.log 2>&1; echo "rc=$?" >> gpurun_out/bench_restore.log
