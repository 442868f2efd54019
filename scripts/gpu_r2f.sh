mkdir -p gpurun_out/r2f
for d in 0 1 2 3; do
  export C2M_CONV_DBG=$d
  echo "== dbg $d" >> gpurun_out/r2f/bench_conv.log
  timeout 120 python scripts/bench_conv.py --only "body" >> gpurun_out/r2f/bench_conv.log 2>&1
  timeout 120 python scripts/bench_conv.py --only "small_offset_conv1" >> gpurun_out/r2f/bench_conv.log 2>&1
done
