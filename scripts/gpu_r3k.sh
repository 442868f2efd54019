set -x
mkdir -p gpurun_out/r3k
O=gpurun_out/r3k
timeout 600 python -m pytest tests/test_conv_gpu.py -q -x -k "autograd or wgrad" 2>&1 | tail -40 > $O/pytest_grad.log
tail -30 $O/pytest_grad.log
timeout 600 python -m pytest tests/test_restoration_gpu.py -q -x -k "training or ddp or stage3 or cfg4" 2>&1 | tail -60 > $O/pytest_train.log
tail -45 $O/pytest_train.log
timeout 300 python bench.py --workload train --steps 10 --warmup 3 > $O/bench_train.log 2>&1; echo "rc=$?" >> $O/bench_train.log
tail -2 $O/bench_train.log | cut -c1-1500
