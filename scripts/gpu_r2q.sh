mkdir -p gpurun_out/r2q
timeout 600 python -m pytest tests/test_conv_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/r2q/pytest.log
timeout 200 python scripts/bench_conv.py --only "body" > gpurun_out/r2q/bench_wino.log 2>&1
timeout 200 python scripts/bench_conv.py --only "offset_conv1" >> gpurun_out/r2q/bench_wino.log 2>&1
