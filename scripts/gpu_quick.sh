mkdir -p gpurun_out/quick
O=gpurun_out/quick
timeout 300 python -m pytest tests/test_dcn_gpu.py -m gpu -q -x -k "forward" 2>&1 | tail -2 > $O/pytest.log
timeout 300 python scripts/bench_dcn.py > $O/bench_dcn.log 2>&1
