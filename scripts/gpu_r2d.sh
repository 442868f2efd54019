set -x
mkdir -p gpurun_out/r2d
for st in 0 1 2 4 9 -1; do
  if [ $st -ge 0 ]; then export C2M_CONV_STAGGER=$st; else unset C2M_CONV_STAGGER; fi
  echo "== stagger $st" >> gpurun_out/r2d/bench_conv.log
  timeout 120 python scripts/bench_conv.py --only "body" >> gpurun_out/r2d/bench_conv.log 2>&1
  timeout 120 python scripts/bench_conv.py --only "small_offset_conv1" >> gpurun_out/r2d/bench_conv.log 2>&1
done
unset C2M_CONV_STAGGER
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_restoration_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r2d/pytest_a.log
