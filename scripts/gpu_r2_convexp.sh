mkdir -p gpurun_out/convexp
O=gpurun_out/convexp
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_restoration_gpu.py -m gpu -q 2>&1 | tail -8 > $O/pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench_default.log 2>&1
C2M_CONV_WINO4=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_nowino4.log 2>&1
