set -x
mkdir -p gpurun_out/dedup
O=gpurun_out/dedup
timeout 900 python -m pytest tests/test_corr_gpu.py tests/test_restoration_gpu.py -m gpu -q -x 2>&1 | tail -15 > $O/pytest.log
timeout 300 python bench.py --workload corr --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_corr.log 2>&1
C2M_CORR_DEDUP=0 timeout 300 python bench.py --workload corr --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_corr_nodedup.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_default.log 2>&1
