mkdir -p gpurun_out/r3m
timeout 600 python -m pytest tests/test_restoration_gpu.py -q -x -k "training_path" 2>&1 | grep -E "^E |Error|assert" | head -20 > gpurun_out/r3m/err.log
cat gpurun_out/r3m/err.log
