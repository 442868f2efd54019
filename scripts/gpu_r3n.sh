set -x
mkdir -p gpurun_out/r3n
O=gpurun_out/r3n
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
tail -5 $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -q -rf 2>&1 | tail -60 > $O/pytest_gpu.log
tail -30 $O/pytest_gpu.log
