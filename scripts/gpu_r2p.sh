mkdir -p gpurun_out/r2p
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_restoration_gpu.py -x -q 2>&1 | tail -25 > gpurun_out/r2p/pytest.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r2p/bench_default.log 2>&1
