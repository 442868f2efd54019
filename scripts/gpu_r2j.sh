mkdir -p gpurun_out/r2j
timeout 900 python -m pytest tests/test_conv_gpu.py -x -q 2>&1 | tail -5 > gpurun_out/r2j/pytest.log
for t in 1 2 4 8; do
  export C2M_CONV_TPW=$t
  echo "== tpw $t" >> gpurun_out/r2j/bench_conv.log
  timeout 120 python scripts/bench_conv.py --only "body" >> gpurun_out/r2j/bench_conv.log 2>&1
  timeout 120 python scripts/bench_conv.py --only "small_offset_conv1" >> gpurun_out/r2j/bench_conv.log 2>&1
done
unset C2M_CONV_TPW
timeout 300 python scripts/bench_conv.py > gpurun_out/r2j/bench_conv_all.log 2>&1
timeout 900 python -m pytest tests/test_restoration_gpu.py -x -q 2>&1 | tail -5 >> gpurun_out/r2j/pytest.log
