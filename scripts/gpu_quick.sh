mkdir -p gpurun_out/quick
O=gpurun_out/quick
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_restoration_gpu.py -m gpu -q -x 2>&1 | tail -4 > $O/pytest.log
timeout 300 python scripts/bench_conv.py --iters 8 2>&1 | grep "^{" > $O/res.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_default.log 2>&1
