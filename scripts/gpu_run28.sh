set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_dcn_gpu.py tests/test_restoration_gpu.py -m gpu -q -rA 2>&1 | tail -30 > gpurun_out/pytest_gpu.log
timeout 900 python scripts/bench_dcn.py > gpurun_out/bench_dcn.log 2>&1
