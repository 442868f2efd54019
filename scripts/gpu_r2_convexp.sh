mkdir -p gpurun_out/convexp
O=gpurun_out/convexp
timeout 300 python scripts/bench_conv.py --iters 8 2>&1 | grep "^{" >> $O/res.log
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_dcn_gpu.py -m gpu -q -x 2>&1 | tail -4 >> $O/res.log
