set -x
mkdir -p gpurun_out/r2b
timeout 600 python -m pytest tests/test_conv_gpu.py -x -q 2>&1 | tail -25 > gpurun_out/r2b/pytest_conv.log
timeout 300 python scripts/bench_conv.py > gpurun_out/r2b/bench_conv.log 2>&1
timeout 1800 python -m pytest tests -m gpu -q --deselect tests/test_conv_gpu.py 2>&1 | tail -30 > gpurun_out/r2b/pytest_gpu.log
