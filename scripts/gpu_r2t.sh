mkdir -p gpurun_out/r2t
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_restoration_gpu.py -x -q 2>&1 | tail -12 > gpurun_out/r2t/pytest.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2t/bench_default.log 2>&1
