mkdir -p gpurun_out/quick
O=gpurun_out/quick
C2M_BENCH_FORCE_DIST=1 timeout 170 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_dist1.log 2>&1
echo "rc=$?" >> $O/bench_dist1.log
