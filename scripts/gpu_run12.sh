set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 > gpurun_out/bench_torchrun.log 2>&1; echo "rc=$?" >> gpurun_out/bench_torchrun.log
timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1; echo "rc=$?" >> gpurun_out/bench_default.log
