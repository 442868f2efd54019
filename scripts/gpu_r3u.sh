for a in 0 1 2 6 8 32 39 47 48; do
echo "ABL $a: $(C2M_SPLIT_ABL=$a timeout 120 python scripts/bench_conv.py --algo split16 --only 'body 64->64 @640' 2>/dev/null | grep "^{'layer" | head -1)"
done
echo "256: $(timeout 120 python scripts/bench_conv.py --algo split16 --only 'small_offset_conv2' 2>/dev/null | grep "^{'layer" | head -1)"
for a in 39 47 48 32; do
echo "256 ABL $a: $(C2M_SPLIT_ABL=$a timeout 120 python scripts/bench_conv.py --algo split16 --only 'small_offset_conv2' 2>/dev/null | grep "^{'layer" | head -1)"
done
