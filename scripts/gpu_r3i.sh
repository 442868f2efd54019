set -x
mkdir -p gpurun_out/r3i
O=gpurun_out/r3i
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_restoration_gpu.py -q -x 2>&1 | tail -15 > $O/pytest.log
tail -4 $O/pytest.log
timeout 400 python bench.py --steps 5 --warmup 2 > $O/bench_split.log 2>&1; echo "rc=$?" >> $O/bench_split.log
C2M_CONV_SPLIT=all timeout 400 python bench.py --steps 5 --warmup 2 > $O/bench_split_all.log 2>&1; echo "rc=$?" >> $O/bench_split_all.log
python - <<'PY'
import json
for f in ("bench_split","bench_split_all"):
    try:
        line=[l for l in open(f"gpurun_out/r3i/{f}.log") if l.startswith("{")][-1]
        d=json.loads(line); print(f, d["value"], d["ms_per_step"], d["stage_ms"]); print(json.dumps(d.get("cpu_baseline"))[:900])
    except Exception as e: print(f, "ERR", e)
PY
