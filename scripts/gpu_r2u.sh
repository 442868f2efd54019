mkdir -p gpurun_out/r2u
timeout 200 python scripts/bench_train.py --graph > gpurun_out/r2u/train_graph.log 2>&1
echo "rc=$?" >> gpurun_out/r2u/train_graph.log
