# round 3, call A: first contact of the split-bf16 convolution kernel with the hardware
set -x
mkdir -p gpurun_out/r3a
O=gpurun_out/r3a
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_gpu.py -q -x -k "split or bf16 or refresh or head_on" 2>&1 | tail -40 > $O/pytest_split.log
cat $O/pytest_split.log | tail -15
timeout 300 python scripts/bench_conv.py --algo split --iters 5 > $O/bench_conv_split.log 2>&1
tail -3 $O/bench_conv_split.log | head -2 | cut -c1-1500
timeout 200 python scripts/bench_conv.py --algo bf16 --iters 5 --only "64 @" > $O/bench_conv_bf16.log 2>&1
C2M_CONV_SPLIT=0 timeout 300 python scripts/bench_conv.py --fast --iters 5 > $O/bench_conv_r2fast.log 2>&1
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_split.log 2>&1; echo "rc=$?" >> $O/bench_split.log
C2M_CONV_SPLIT=0 timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_r2.log 2>&1; echo "rc=$?" >> $O/bench_r2.log
C2M_CONV_SPLIT=all timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_split_all.log 2>&1; echo "rc=$?" >> $O/bench_split_all.log
python - <<'PY'
import json
for f in ("bench_split","bench_r2","bench_split_all"):
    try:
        line=[l for l in open(f"gpurun_out/r3a/{f}.log") if l.startswith("{")][-1]
        d=json.loads(line); print(f, d["value"], d["ms_per_step"], d["stage_ms"])
    except Exception as e: print(f, "ERR", e)
PY
