set -x
mkdir -p gpurun_out/r3s
O=gpurun_out/r3s
timeout 1200 python -m pytest tests/test_conv_gpu.py -q -x -k "split or refresh or head" 2>&1 | tail -15 > $O/pytest_conv.log
tail -15 $O/pytest_conv.log
for a in split split16; do
timeout 300 python scripts/bench_conv.py --algo $a 2>&1 | tail -6
done
