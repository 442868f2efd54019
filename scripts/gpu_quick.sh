mkdir -p gpurun_out/quick
O=gpurun_out/quick
timeout 900 python -m pytest tests/test_dcn_gpu.py tests/test_restoration_gpu.py -m gpu -q -x 2>&1 | tail -6 > $O/pytest.log
timeout 600 python scripts/bench_dcn.py > $O/bench_dcn.log 2>&1
C2M_DCN_BWD_GROUPED=0 timeout 600 python scripts/bench_dcn.py > $O/bench_dcn_old.log 2>&1
timeout 600 python scripts/bench_dcn.py --bwd-batch 16 --bwd-lr 160 > $O/bench_dcn_big.log 2>&1
C2M_DCN_BWD_GROUPED=0 timeout 600 python scripts/bench_dcn.py --bwd-batch 16 --bwd-lr 160 > $O/bench_dcn_big_old.log 2>&1
