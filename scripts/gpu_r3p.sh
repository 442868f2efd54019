set -x
mkdir -p gpurun_out/r3p
O=gpurun_out/r3p
timeout 600 python -m pytest tests/test_conv_gpu.py -q -x -k "split or bf16 or refresh or head_on or autograd or wgrad" 2>&1 | tail -12 > $O/pytest_split.log
tail -5 $O/pytest_split.log
timeout 300 python scripts/bench_conv.py --algo split --iters 5 > $O/bench_conv_split.log 2>&1
grep "^{'layer" $O/bench_conv_split.log
timeout 200 python scripts/bench_conv.py --algo bf16 --iters 5 --only "64 @" 2>&1 | grep "^{'layer"
python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value'],2), round(d['ms_per_step'],2), {k: round(v,1) for k,v in d['stage_ms'].items()})"
