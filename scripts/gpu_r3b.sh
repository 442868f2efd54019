# round 3, call C: remaining split tests + timing ablations of the split kernel on the 64->64 @640 body layer
set -x
mkdir -p gpurun_out/r3c
O=gpurun_out/r3c
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_gpu.py -q -k "split or bf16 or refresh or head_on" 2>&1 | tail -40 > $O/pytest_split.log
tail -5 $O/pytest_split.log
for a in 0 1 2 3 4 5 6; do
  echo "ABL=$a" >> $O/abl.log
  C2M_SPLIT_ABL=$a timeout 120 python scripts/bench_conv.py --algo split --iters 5 --only "body 64->64 @640" 2>&1 | grep "^{'layer" >> $O/abl.log
  C2M_SPLIT_ABL=$a timeout 120 python scripts/bench_conv.py --algo split --iters 5 --only "large_offset_conv1" 2>&1 | grep "^{'layer" >> $O/abl.log
done
for t in 1 2 4 10; do
  echo "TPW=$t" >> $O/abl.log
  C2M_CONV_TPW=$t timeout 120 python scripts/bench_conv.py --algo split --iters 5 --only "body 64->64 @640" 2>&1 | grep "^{'layer" >> $O/abl.log
done
cat $O/abl.log
