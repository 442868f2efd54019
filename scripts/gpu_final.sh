# Round-end validation + measurement on one MI355X: everything DESIGN.md / profiles/ quote comes from this script.
set -x
mkdir -p gpurun_out/final
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -q -rA 2>&1 | tail -170 > $O/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.log 2>&1; echo "rc=$?" >> $O/bench_default.log
timeout 300 python bench.py --workload corr --steps 10 --warmup 3 > $O/bench_corr.log 2>&1
timeout 300 python bench.py --workload train --steps 20 --warmup 5 > $O/bench_train.log 2>&1
timeout 300 python bench.py --workload train --lr 96 --steps 8 --warmup 3 > $O/bench_train_lr96.log 2>&1
timeout 300 python bench.py --workload train --steps 20 --warmup 5 --graph 1 > $O/bench_train_hipgraph.log 2>&1
C2M_BENCH_FORCE_DIST=1 timeout 300 python bench.py --workload train --steps 10 --warmup 3 > $O/bench_train_rccl_1rank.log 2>&1
timeout 300 python bench.py --lr 320 --dtype bf16 --steps 5 --warmup 2 > $O/bench_cfg5_bf16.log 2>&1
timeout 300 python bench.py --lr 320 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg5_f32.log 2>&1
timeout 300 python scripts/bench_conv.py > $O/bench_conv.log 2>&1
echo "--- C2M_CONV_SPLIT=0 --fast (round-2 kernels: Winograd F(4,3) / F(2,3) / direct on fp32 MFMA)" >> $O/bench_conv.log
C2M_CONV_SPLIT=0 timeout 300 python scripts/bench_conv.py --fast >> $O/bench_conv.log 2>&1
echo "--- --algo bf16 (single-piece flavour, configs[4])" >> $O/bench_conv.log
timeout 300 python scripts/bench_conv.py --algo bf16 >> $O/bench_conv.log 2>&1
echo "--- --algo split (bf16 x 3 flavour: six products; autograd path and C2M_CONV_SPLIT16=0)" >> $O/bench_conv.log
timeout 300 python scripts/bench_conv.py --algo split >> $O/bench_conv.log 2>&1
# power experiment: the same instruction stream on N(0,1) and on all-zero tensors; MFMA-only ablation (mask 111) of both
for d in randn zeros ones randn; do echo "split16 $d: $(timeout 120 python scripts/bench_conv.py --algo split16 --only 'body 64->64 @640' --iters 20 --data $d 2>/dev/null | grep "^{'layer" | head -1)"; done > $O/power_experiment.log
for d in randn zeros; do echo "split16 MFMA-only (C2M_SPLIT_ABL=111) $d: $(C2M_SPLIT_ABL=111 timeout 120 python scripts/bench_conv.py --algo split16 --only 'body 64->64 @640' --iters 20 --data $d 2>/dev/null | grep "^{'layer" | head -1)"; done >> $O/power_experiment.log
for d in randn zeros; do echo "split (bf16 x 3) $d: $(timeout 120 python scripts/bench_conv.py --algo split --only 'body 64->64 @640' --iters 20 --data $d 2>/dev/null | grep "^{'layer" | head -1)"; done >> $O/power_experiment.log
timeout 600 python scripts/bench_dcn.py > $O/bench_dcn.log 2>&1
timeout 120 scripts/ubench/mfma_bf16_rate > $O/ubench_mfma_bf16_rate.log 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_step -o step -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-alt > $O/rocprof_step.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace -f csv -d $O/pmc_mfma -o step -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-alt > $O/pmc_mfma.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $O/pmc_fetch -o step -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-alt > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $O/pmc_write -o step -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-alt > $O/pmc_write.log 2>&1
cd $R
for f in $(find gpurun_out/final -name "*.db"); do rm -f $f; done
# the per-dispatch traces are large: keep only the stats / counter CSVs; then: python scripts/summarize_step.py r03_final 4
find gpurun_out/final -name "*kernel_trace.csv" -size +8M -delete
