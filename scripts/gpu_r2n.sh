mkdir -p gpurun_out/r2n
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_data_path.py tests/test_metrics.py -x -q 2>&1 | tail -5 > gpurun_out/r2n/pytest.log
timeout 300 python scripts/bench_conv.py > gpurun_out/r2n/bench_conv_all.log 2>&1
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2n/bench_default.log 2>&1
