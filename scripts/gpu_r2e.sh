set -x
mkdir -p gpurun_out/r2e
timeout 600 python -m pytest tests/test_conv_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/r2e/pytest_conv.log
for st in 0 2 4 9; do
  export C2M_CONV_STAGGER=$st
  echo "== stagger $st" >> gpurun_out/r2e/bench_conv.log
  timeout 120 python scripts/bench_conv.py --only "body" >> gpurun_out/r2e/bench_conv.log 2>&1
  timeout 120 python scripts/bench_conv.py --only "small_offset_conv1" >> gpurun_out/r2e/bench_conv.log 2>&1
done
unset C2M_CONV_STAGGER
timeout 300 python scripts/bench_conv.py > gpurun_out/r2e/bench_conv_all.log 2>&1
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2e/bench_default.log 2>&1
