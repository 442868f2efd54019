set -x
mkdir -p gpurun_out/r3j
O=gpurun_out/r3j
timeout 900 python -m pytest tests/test_restoration_gpu.py -q -x -k "cfg5 or ddp or dataparallel or bf16" 2>&1 | tail -25 > $O/pytest_new.log
tail -6 $O/pytest_new.log
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench_default.log 2>&1; echo "rc=$?" >> $O/bench_default.log
timeout 300 python bench.py --workload train --steps 10 --warmup 3 > $O/bench_train.log 2>&1; echo "rc=$?" >> $O/bench_train.log
timeout 300 python bench.py --lr 320 --dtype bf16 --steps 3 --warmup 1 > $O/bench_cfg5_bf16.log 2>&1; echo "rc=$?" >> $O/bench_cfg5_bf16.log
timeout 300 python bench.py --lr 320 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cfg5_f32.log 2>&1; echo "rc=$?" >> $O/bench_cfg5_f32.log
C2M_BENCH_FORCE_DIST=1 timeout 300 python bench.py --workload train --steps 5 --warmup 2 > $O/bench_train_ddp1.log 2>&1; echo "rc=$?" >> $O/bench_train_ddp1.log
for f in bench_default bench_train bench_cfg5_bf16 bench_cfg5_f32 bench_train_ddp1; do echo "== $f"; tail -3 $O/$f.log | cut -c1-700; done
