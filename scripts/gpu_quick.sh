mkdir -p gpurun_out/quick
O=gpurun_out/quick
timeout 600 python -m pytest tests/test_conv_gpu.py -m gpu -q -x -k "f43" 2>&1 | tail -3 > $O/pytest.log
timeout 300 python scripts/bench_conv.py --only "64 @" --iters 10 --fast 2>&1 | grep "^{" > $O/res.log
