set -x
mkdir -p gpurun_out/r3t
O=gpurun_out/r3t
python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tee $O/bench.log | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value'],2), round(d['ms_per_step'],2), {k: round(v,1) for k,v in d['stage_ms'].items()}); print([(r['kernel'][:24], round(r['kernel_ms'],1), round(r['frac'],3)) for r in d['roofline_kernels']])"
timeout 2400 python -m pytest tests -q -x -m gpu 2>&1 | tail -8 > $O/pytest_gpu.log
tail -8 $O/pytest_gpu.log
