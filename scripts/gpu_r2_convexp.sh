mkdir -p gpurun_out/convexp
O=gpurun_out/convexp
timeout 600 python -m pytest tests/test_conv_gpu.py -m gpu -q -x -k "head" 2>&1 | tail -5 > $O/pytest.log
timeout 300 python scripts/bench_conv.py --only "head" --iters 8 2>&1 | grep "^{" > $O/res.log
