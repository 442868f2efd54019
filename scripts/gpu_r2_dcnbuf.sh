set -x
mkdir -p gpurun_out/dcnbuf
O=gpurun_out/dcnbuf
timeout 900 python -m pytest tests/test_dcn_gpu.py tests/test_conv_gpu.py tests/test_restoration_gpu.py -m gpu -q -x 2>&1 | tail -8 > $O/pytest.log
timeout 600 python scripts/bench_dcn.py > $O/bench_dcn.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_default.log 2>&1
