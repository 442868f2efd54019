mkdir -p gpurun_out/r2s
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_restoration_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/r2s/pytest.log
timeout 300 python scripts/bench_conv.py > gpurun_out/r2s/bench_conv.log 2>&1
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2s/bench_default.log 2>&1
