# quick validation on one MI355X: smoke + the GPU test suite (the full measurement pass is scripts/gpu_final.sh)
mkdir -p gpurun_out/quick
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/quick/smoke.log 2>&1; tail -2 gpurun_out/quick/smoke.log
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/quick/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/quick/pytest_gpu.log | tail -3
