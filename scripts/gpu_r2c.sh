set -x
mkdir -p gpurun_out/r2c
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_restoration_gpu.py -x -q 2>&1 | tail -40 > gpurun_out/r2c/pytest_a.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r2c/bench_default.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_conv_gpu.py --deselect tests/test_restoration_gpu.py 2>&1 | tail -15 > gpurun_out/r2c/pytest_b.log
