set -x
mkdir -p gpurun_out/r3o
O=gpurun_out/r3o
run() { python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value'],2), round(d['ms_per_step'],2), {k: round(v,1) for k,v in d['stage_ms'].items()})"; }
echo "default" >> $O/t.log; run >> $O/t.log 2>&1
echo "C2M_WEIGHT_REFRESH=0" >> $O/t.log; C2M_WEIGHT_REFRESH=0 run >> $O/t.log 2>&1
for t in 1 2 3 5 8; do echo "C2M_CONV_TPW=$t" >> $O/t.log; C2M_CONV_TPW=$t run >> $O/t.log 2>&1; done
echo "default again" >> $O/t.log; run >> $O/t.log 2>&1
cat $O/t.log
