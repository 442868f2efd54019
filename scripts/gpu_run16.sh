set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_corr_gpu.py -m gpu -q -rA 2>&1 | tail -30 > gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench.log 2>&1
timeout 600 python scripts/corr_ablation.py > gpurun_out/corr_ablation.log 2>&1
