set -x
mkdir -p gpurun_out/r3g
O=gpurun_out/r3g
for a in 0 7 8 5 3; do
  echo "ABL=$a" >> $O/abl.log
  C2M_SPLIT_ABL=$a timeout 60 python scripts/bench_conv.py --algo split --iters 5 --only "body 64->64 @640" > $O/abl_$a.log 2>&1
  grep "^{'layer" $O/abl_$a.log >> $O/abl.log; tail -2 $O/abl_$a.log | cut -c1-300 >> $O/abl.log
done
cat $O/abl.log
