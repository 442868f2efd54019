set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -8 > gpurun_out/rocminfo.txt 2>&1
nproc > gpurun_out/nproc.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> gpurun_out/nproc.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 900 python -m pytest tests -m gpu -q -rA 2>&1 | tail -60 > gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_corr -o corr -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/rocprof.log 2>&1
cd /root/repo; ls -R gpurun_out | head -50
