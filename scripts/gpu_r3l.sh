set -x
mkdir -p gpurun_out/r3l
O=gpurun_out/r3l
timeout 600 python -m pytest tests/test_restoration_gpu.py tests/test_conv_gpu.py -q -x -k "training_path or wgrad or autograd" 2>&1 | tail -8 > $O/pytest.log
tail -4 $O/pytest.log
for tk in 0 1; do
 for cfg in "--lr 40 --batch 4" "--lr 96 --batch 4" "--lr 160 --batch 4"; do
  echo "C2M_TRAIN_KERNELS=$tk $cfg" >> $O/train.log
  C2M_TRAIN_KERNELS=$tk timeout 300 python bench.py --workload train --steps 8 --warmup 3 $cfg 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value'],1), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['c2m_kernel_ms_per_step'].items()})" >> $O/train.log 2>&1
 done
done
cat $O/train.log
