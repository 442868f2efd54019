mkdir -p gpurun_out/r3v
timeout 1200 python -m pytest tests/test_conv_gpu.py -q -x -k "split or refresh or head or bf16" 2>&1 | tail -6
for a in 0 2 32 39 47 48; do
echo "ABL $a: $(C2M_SPLIT_ABL=$a timeout 120 python scripts/bench_conv.py --algo split16 --only 'body 64->64 @640' 2>/dev/null | grep "^{'layer" | head -1)"
done
timeout 300 python scripts/bench_conv.py --algo split16 2>&1 | grep "^{'layer"
