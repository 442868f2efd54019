set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rA 2>&1 | tail -45 > gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench.log 2>&1
timeout 900 python scripts/bench_train.py > gpurun_out/bench_train.log 2>&1
