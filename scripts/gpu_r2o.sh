mkdir -p gpurun_out/r2o
timeout 600 python -m pytest tests/test_conv_gpu.py -x -q 2>&1 | tail -25 > gpurun_out/r2o/pytest.log
timeout 200 python scripts/bench_conv.py --only "body" > gpurun_out/r2o/bench_wino.log 2>&1
C2M_CONV_WINO=0 timeout 200 python scripts/bench_conv.py --only "body" > gpurun_out/r2o/bench_direct.log 2>&1
