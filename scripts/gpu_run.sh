#!/bin/bash
# One parametrised GPU lease script (replaces the per-experiment scripts of earlier rounds):
#   gpurun --timeout N -- 'bash scripts/gpu_run.sh <tag> <stage> [<stage> ...]'
# Stages write under gpurun_out/<tag>/.  Everything DESIGN.md / profiles/ quote comes from `final`.
set -x
TAG=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
for stage in "$@"; do
  case $stage in
    smoke)      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log ;;
    diag_w16)   (timeout 300 python scripts/diag_wino16.py; timeout 300 python scripts/diag_wino16.py f23) > $O/diag_wino16.log 2>&1 ;;
    bisect_w16) (for m in 0 1 2 4 3 7; do echo "=== C2M_W16_DBG=$m"; C2M_W16_DBG=$m timeout 120 python scripts/diag_wino16.py 0 2 3 2>&1 | grep -v amdgpu.ids | cut -c1-400; done) > $O/bisect_wino16.log 2>&1 ;;
    abl_w16)    (for m in ${W16_MASKS:-0 1 2 4 8 16 32 64 3 19 27 59 72 123}; do echo "=== C2M_W16_DBG=$m (1 no re-loads, 2 no items, 4 no weight DMA, 8 no unit-end waits/barriers, 16 one output row of four stored, 32 no MFMAs)"; C2M_W16_DBG=$m timeout 120 python scripts/bench_conv.py --algo ${W16_ALGO:-wino16} --only "64->64 @640" --iters 20 2>&1 | grep "^{'layer"; done) > $O/abl_wino16.log 2>&1 ;;
    abl128)     (for abl in 0 128; do echo "=== C2M_SPLIT_ABL=$abl (128: the first unit-end wait after a tile's epilogue lets its 16 stores stay in flight)"; C2M_SPLIT_ABL=$abl timeout 200 python scripts/bench_conv.py --algo split16 --only "64->64" --iters 20 2>&1 | grep "^{'layer"; done
                 echo "=== conv tests under C2M_SPLIT_ABL=128"; C2M_SPLIT_ABL=128 timeout 600 python -m pytest tests/test_conv_gpu.py -m gpu -q -x -k "split16 and (fp64 or full_size or scales)" 2>&1 | tail -15) > $O/abl128.log 2>&1 ;;
    pmc_w16)    cd /tmp
                for a in split16 wino16 wino16_f23; do
                  timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_LDS --kernel-trace -f csv -d $O/pmcw_$a -o c -- python $R/scripts/bench_conv.py --algo $a --only 'body 64->64 @640' --iters 6 > $O/pmcw_$a.log 2>&1
                  echo "=== --algo $a" >> $O/pmc_wino16.txt
                  grep "^{'layer" $O/pmcw_$a.log >> $O/pmc_wino16.txt
                  python $R/scripts/pmc_kernel.py $O/pmcw_$a conv3x3_ >> $O/pmc_wino16.txt 2>&1
                  timeout 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS --kernel-trace -f csv -d $O/pmcx_$a -o c -- python $R/scripts/bench_conv.py --algo $a --only 'body 64->64 @640' --iters 6 > $O/pmcx_$a.log 2>&1
                  python $R/scripts/pmc_kernel.py $O/pmcx_$a conv3x3_ >> $O/pmc_wino16.txt 2>&1
                  rm -rf $O/pmcw_$a $O/pmcx_$a
                done
                cd $R ;;
    ab_bf)      (for lib in "" $R/build_exp/${AB_LIB:-libc2m_base.so} "" $R/build_exp/${AB_LIB:-libc2m_base.so}; do echo "=== C2M_LIB=$lib"; C2M_LIB=$lib timeout 300 python scripts/bench_conv.py --algo split16 --iters 20 2>&1 | grep "^{'layer"; done) > $O/ab_branch_free.log 2>&1
                (echo "=== in-tree (branch-free chunk loop)"; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt 2>&1 | grep "^{" | cut -c1-1600
                 echo "=== C2M_LIB=build_exp/${AB_LIB:-libc2m_base.so}"; C2M_LIB=$R/build_exp/${AB_LIB:-libc2m_base.so} timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt 2>&1 | grep "^{" | cut -c1-1600) > $O/ab_branch_free_step.log 2>&1 ;;
    abl_corrf)  (for lib in "" ${CF_LIBS:-cf1 cf2 cf4 cf8 cf16 cf3 cf15} ""; do echo "=== ${lib:-in-tree} (C2M_CORRF_ABL: 1 B operands reused, 2 no ring reads, 4 no tap rounds, 8 no row-sum tail, 16 no MFMAs)"; C2M_LIB=${lib:+$R/build_exp/libc2m_$lib.so} timeout 120 python scripts/abl_corr_filter.py 2>&1 | grep "^{"; done) > $O/abl_corr_filter.log 2>&1 ;;
    abl_c3)     (for lib in "" ${C3_LIBS:-c3a1 c3a2 c3a4 c3a3 c3a7} ""; do echo "=== ${lib:-in-tree} (C2M_C3_ABL: 1 one store of eight, 2 no MFMAs, 4 tile staged once)"; C2M_LIB=${lib:+$R/build_exp/libc2m_$lib.so} timeout 120 python scripts/abl_c3.py 2>&1 | grep "^{"; done) > $O/abl_c3.log 2>&1 ;;
    test_w16)   timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q -rA -k "wino16" 2>&1 | tail -120 > $O/pytest_wino16.log ;;
    bench_w16)  (for a in split16 wino16 wino16_f23; do echo "== $a"; timeout 200 python scripts/bench_conv.py --algo $a --only "body" --iters 20; done) 2>&1 | grep -v "^\[{" > $O/bench_wino16.log ;;
    test_corr)  timeout 900 python -m pytest tests/test_corr_gpu.py -m gpu -q -rA 2>&1 | tail -80 > $O/pytest_corr.log ;;
    test_conv)  timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q -rA 2>&1 | tail -80 > $O/pytest_conv.log ;;
    test_dcn)   timeout 900 python -m pytest tests/test_dcn_gpu.py -m gpu -q -rA 2>&1 | tail -80 > $O/pytest_dcn.log ;;
    test_rest)  timeout 1500 python -m pytest tests/test_restoration_gpu.py -m gpu -q -rA 2>&1 | tail -80 > $O/pytest_restoration.log ;;
    test_all)   timeout 2400 python -m pytest tests -m gpu -q -rA 2>&1 | tail -220 > $O/pytest_gpu.log ;;
    diag_corr)  timeout 600 python scripts/diag_corr_filter.py --lr320 > $O/diag_corr_filter.log 2>&1 ;;
    bench)      timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.log 2>&1; echo "rc=$?" >> $O/bench_default.log ;;
    bench_dist1) C2M_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-alt > $O/bench_dist1.log 2>&1; echo "rc=$?" >> $O/bench_dist1.log ;;
    bench_quick) timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-alt > $O/bench_quick.log 2>&1; echo "rc=$?" >> $O/bench_quick.log ;;
    bench_corr) timeout 300 python bench.py --workload corr --steps 10 --warmup 3 > $O/bench_corr.log 2>&1 ;;
    bench_train) timeout 300 python bench.py --workload train --steps 20 --warmup 5 > $O/bench_train.log 2>&1
                C2M_TRAIN_KERNELS=0 timeout 300 python bench.py --workload train --steps 20 --warmup 5 > $O/bench_train_stock.log 2>&1
                timeout 300 python bench.py --workload train --batch 4 --steps 20 --warmup 5 > $O/bench_train_b4.log 2>&1
                C2M_TRAIN_KERNELS=0 timeout 300 python bench.py --workload train --batch 4 --steps 20 --warmup 5 > $O/bench_train_b4_stock.log 2>&1
                C2M_TRAIN_KERNELS=1 timeout 300 python bench.py --workload train --steps 20 --warmup 5 > $O/bench_train_kernels.log 2>&1
                C2M_BENCH_FORCE_DIST=1 timeout 300 python bench.py --workload train --steps 10 --warmup 3 > $O/bench_train_rccl_1rank.log 2>&1 ;;
    bench_cfg5) timeout 300 python bench.py --lr 320 --dtype bf16 --steps 5 --warmup 2 > $O/bench_cfg5_bf16.log 2>&1
                timeout 300 python bench.py --lr 320 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg5_f32.log 2>&1 ;;
    bench_cfg5_bf16) timeout 300 python bench.py --lr 320 --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg5_bf16.log 2>&1
                C2M_BF16_IO=0 timeout 300 python bench.py --lr 320 --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg5_bf16_f32io.log 2>&1 ;;
    bench_conv16) (echo "== bf16 kernel, fp32 tensors, B=4"; timeout 200 python scripts/bench_conv.py --batch 4 --algo bf16 --only body
                   echo "== bf16 kernel, bf16 tensors, B=4"; timeout 200 python scripts/bench_conv.py --batch 4 --io16 --only body) > $O/bench_conv16.log 2>&1 ;;
    abl16)      for abl in 0 1 2 8 16 32 64 43 48 107; do
                  echo "=== C2M_SPLIT_ABL16=$abl (1 no weight DMA, 2 no halo DMA, 8 no waits/barriers, 16 no MFMAs, 32 no operand reads, 64 one store)" >> $O/abl16.txt
                  C2M_SPLIT_ABL16=$abl timeout 200 python scripts/bench_conv.py --batch 4 --io16 --only '64->64 @1280' --iters 20 2>&1 | grep "^{'layer" >> $O/abl16.txt
                done ;;
    abl16_pmc)  cd /tmp
                for abl in 0 107 16 43; do
                  C2M_SPLIT_ABL16=$abl timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_LDS --kernel-trace -f csv -d $O/abl16_$abl -o c -- python $R/scripts/bench_conv.py --batch 4 --io16 --only 'body 64->64 @1280' --iters 6 > $O/abl16_$abl.log 2>&1
                  echo "=== C2M_SPLIT_ABL16=$abl" >> $O/abl16_pmc.txt
                  python $R/scripts/pmc_kernel.py $O/abl16_$abl conv3x3_split_kernel >> $O/abl16_pmc.txt 2>&1
                  C2M_SPLIT_ABL16=$abl timeout 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS --kernel-trace -f csv -d $O/abl16b_$abl -o c -- python $R/scripts/bench_conv.py --batch 4 --io16 --only 'body 64->64 @1280' --iters 6 > $O/abl16b_$abl.log 2>&1
                  python $R/scripts/pmc_kernel.py $O/abl16b_$abl conv3x3_split_kernel >> $O/abl16_pmc.txt 2>&1
                  rm -rf $O/abl16_$abl $O/abl16b_$abl
                done
                cd $R ;;
    test_head)  timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q -rA -k "head" 2>&1 | tail -60 > $O/pytest_head.log ;;
    bench_head) (timeout 200 python scripts/bench_conv.py --only "dcn head"; echo "== C2M_HEAD_QUAD=0"; C2M_HEAD_QUAD=0 timeout 200 python scripts/bench_conv.py --only "dcn head") > $O/bench_head.log 2>&1 ;;
    test_dcn16) timeout 900 python -m pytest tests/test_dcn_gpu.py -m gpu -q -x -k "f16x2" 2>&1 | tail -60 > $O/pytest_dcn16.log ;;
    bench_dcn16) timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-alt > $O/bench_dcn16_on.log 2>&1
                C2M_DCN_F16X2=0 timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-alt > $O/bench_dcn16_off.log 2>&1 ;;
    dcn_variants) (echo "== in-tree"; timeout 300 python scripts/bench_dcn_nhwc.py
                   for v in $DCN_VARIANTS; do echo "== $v"; C2M_LIB=$R/build_exp/libc2m_$v.so timeout 300 python scripts/bench_dcn_nhwc.py; done) 2>&1 | grep -v "^\[{" > $O/dcn_variants.log ;;
    abl512)     for abl in 0 512 64; do
                  echo "=== C2M_SPLIT_ABL=$abl" >> $O/abl512.txt
                  C2M_SPLIT_ABL=$abl timeout 200 python scripts/bench_conv.py --algo split16 --only '64->64 @640' --iters 20 2>&1 | grep "^{'layer" >> $O/abl512.txt
                done ;;
    tpw_sweep)  for t in 0 1 2 3 4 5 6 7 9 13; do
                  echo "=== C2M_CONV_TPW=$t (0 = heuristic)" >> $O/tpw_sweep.txt
                  C2M_CONV_TPW=$t timeout 200 python scripts/bench_conv.py --only "${TPW_ONLY:-64->64 @320}" --iters 20 2>&1 | grep "^{'layer" >> $O/tpw_sweep.txt
                done ;;
    test_bf16)  timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_restoration_gpu.py -m gpu -q -rA -k "bf16" 2>&1 | tail -80 > $O/pytest_bf16.log ;;
    bench_conv) timeout 300 python scripts/bench_conv.py > $O/bench_conv.log 2>&1 ;;
    bench_dcn)  timeout 600 python scripts/bench_dcn.py > $O/bench_dcn.log 2>&1 ;;
    prof)       cd /tmp
                timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_step -o step -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-alt > $O/rocprof_step.log 2>&1
                cd $R ;;
    ab_mix)     (cd scripts/ubench && timeout 120 ./split_mix_check) > $O/split_mix_check.log 2>&1
                (timeout 300 python scripts/diag_split_bits.py > $O/bits_mix.log 2>&1; C2M_LIB=$R/build_exp/nomix/libc2m_hip.so timeout 300 python scripts/diag_split_bits.py > $O/bits_nomix.log 2>&1
                 grep "^{" $O/bits_mix.log > $O/bits_a.txt; grep "^{" $O/bits_nomix.log > $O/bits_b.txt
                 echo "lines: $(wc -l < $O/bits_a.txt) / $(wc -l < $O/bits_b.txt); differing lines: $(diff $O/bits_a.txt $O/bits_b.txt | grep -c '^<')") > $O/bits_diff.txt 2>&1
                (for lib in "" $R/build_exp/nomix/libc2m_hip.so "" $R/build_exp/nomix/libc2m_hip.so; do echo "=== C2M_LIB=$lib"; C2M_LIB=$lib timeout 300 python scripts/bench_conv.py --algo split16 --iters 20 2>&1 | grep "^{'layer"; done) > $O/ab_mix_layers.log 2>&1
                (for lib in "" $R/build_exp/nomix/libc2m_hip.so "" $R/build_exp/nomix/libc2m_hip.so; do echo "=== C2M_LIB=$lib"; C2M_LIB=$lib timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt 2>&1 | grep "^{" | cut -c1-700; done) > $O/ab_mix_step.log 2>&1 ;;
    ub_ta)      (cd scripts/ubench && timeout 300 ./vmem_ta_cost) > $O/ubench_vmem_ta_cost.log 2>&1
                cd /tmp
                for fp in 0 1; do
                  timeout 300 rocprofv3 --pmc TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace -f csv -d $O/pmc_ubta$fp -o c -- $R/scripts/ubench/vmem_ta_cost $fp > $O/pmc_ubta$fp.log 2>&1
                  echo "=== footprint $fp (0 = L2, 1 = HBM)" >> $O/ubench_vmem_ta_pmc.txt
                  python $R/scripts/pmc_kernel.py $O/pmc_ubta$fp "void k<" >> $O/ubench_vmem_ta_pmc.txt 2>&1
                  rm -rf $O/pmc_ubta$fp
                done
                cd $R ;;
    abl512)     (for abl in 0 512 0 512; do echo "=== C2M_SPLIT_ABL=$abl (512: quad address pattern for stores and residual loads -- lanes 4q..4q+3 cover 64 contiguous bytes)"; C2M_SPLIT_ABL=$abl timeout 200 python scripts/bench_conv.py --algo split16 --only "64->64" --iters 20 2>&1 | grep "^{'layer"; done) > $O/abl512.log 2>&1 ;;
    trace)      (C2M_LIB=$R/build_exp/trace/libc2m_hip.so C2M_SPLIT_ABL=1024 C2M_SPLIT_TRACE_IT=4 timeout 300 python scripts/trace_split.py 640 16
                 C2M_LIB=$R/build_exp/trace/libc2m_hip.so C2M_SPLIT_ABL=1024 C2M_SPLIT_TRACE_IT=1 timeout 300 python scripts/trace_split.py 160 16
                 for lib in "" $R/build_exp/trace/libc2m_hip.so; do echo "=== timing, C2M_LIB=$lib"; C2M_LIB=$lib C2M_SPLIT_ABL=${lib:+1024} timeout 200 python scripts/bench_conv.py --algo split16 --only "64->64 @640" --iters 20 2>&1 | grep "^{'layer"; done) > $O/trace_split.log 2>&1 ;;
    power)      (rocm-smi -M 2>&1 | grep -i "power\|GPU"; rocm-smi -P -c -t --json 2>&1 | cut -c1-900
                 BC="python $R/scripts/bench_conv.py --algo split16 --iters 6000"
                 bash scripts/power_probe.sh "$BC --only 'body 64->64 @640'" "f16 x 2 body 64->64 @640, N(0,1) data"
                 bash scripts/power_probe.sh "$BC --only 'body+res 64->64 @640'" "f16 x 2 body+res 64->64 @640, N(0,1) data"
                 bash scripts/power_probe.sh "$BC --only 'body 64->64 @640' --data zeros" "f16 x 2 body 64->64 @640, all-zero data"
                 C2M_SPLIT_ABL=48 bash scripts/power_probe.sh "$BC --only 'body 64->64 @640'" "the same without its MFMAs and operand reads (C2M_SPLIT_ABL=48)"
                 C2M_SPLIT_ABL=47 bash scripts/power_probe.sh "$BC --only 'body 64->64 @640'" "MFMAs only: no loads, split, operand reads, barriers (C2M_SPLIT_ABL=47)") > $O/power_probe.log 2>&1 ;;
    power2)     (C2M_SPLIT_ABL=32 bash scripts/power_probe.sh "python $R/scripts/bench_conv.py --algo split16 --iters 6000 --only 'body 64->64 @640'" "f16 x 2 body 64->64 @640 without the operand ds_reads (C2M_SPLIT_ABL=32)"
                 bash scripts/power_probe.sh "python $R/scripts/bench_conv.py --algo bf16 --io16 --iters 6000 --only 'body 64->64 @640'" "bf16 tensors, one product: body 64->64 @640"
                 bash scripts/power_probe.sh "python $R/bench.py --workload corr --steps 400 --warmup 3 --no-cpu-baseline" "correlation stage alone (bench.py --workload corr)" 12 8
                 bash scripts/power_probe.sh "python $R/scripts/bench_dcn_nhwc.py --iters 400" "DCNv2 forwards (three layers, fp32 and f16 x 2 in turn)" 16 6
                 bash scripts/power_probe.sh "python $R/bench.py --steps 120 --warmup 3 --no-cpu-baseline --no-alt" "the whole configs[2] step, back to back" 40 10) > $O/power_probe2.log 2>&1 ;;
    abl2048)    (for abl in 0 2048 32 0 2048; do echo "=== C2M_SPLIT_ABL=$abl (2048: 12 of a chunk's 36 B-operand reads skipped -- the re-reads of a pixel row the other accumulator row read one kernel row earlier; 32: no operand reads at all)"; C2M_LIB=$R/build_exp/trace/libc2m_hip.so C2M_SPLIT_ABL=$abl timeout 200 python scripts/bench_conv.py --algo split16 --only "64->64" --iters 20 2>&1 | grep "^{'layer"; done
                 C2M_SPLIT_ABL=2048 bash scripts/power_probe.sh "C2M_LIB=$R/build_exp/trace/libc2m_hip.so python $R/scripts/bench_conv.py --algo split16 --iters 6000 --only 'body 64->64 @640'" "f16 x 2 body 64->64 @640, 12 of 36 B reads skipped (C2M_SPLIT_ABL=2048)") > $O/abl2048.log 2>&1 ;;
    ab_epi)     (timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q -x 2>&1 | tail -5) > $O/ab_epi_tests.log 2>&1
                (for lib in "" $R/build_exp/nomix/libc2m_hip.so "" $R/build_exp/nomix/libc2m_hip.so; do echo "=== C2M_LIB=$lib"; C2M_LIB=$lib timeout 300 python scripts/bench_conv.py --algo split16 --iters 20 2>&1 | grep "^{'layer"; done) > $O/ab_epi_layers.log 2>&1
                (for lib in "" $R/build_exp/nomix/libc2m_hip.so "" $R/build_exp/nomix/libc2m_hip.so; do echo "=== C2M_LIB=$lib"; C2M_LIB=$lib timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt 2>&1 | grep "^{" | cut -c1-700; done) > $O/ab_epi_step.log 2>&1 ;;
    ab_epi16)   (timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_restoration_gpu.py -m gpu -q -x -k "bf16 or io16 or cfg5 or autocast" 2>&1 | tail -5) > $O/ab_epi16_tests.log 2>&1
                (for lib in "" $R/build_exp/nomix/libc2m_hip.so "" $R/build_exp/nomix/libc2m_hip.so; do echo "=== C2M_LIB=$lib"; C2M_LIB=$lib timeout 300 python scripts/bench_conv.py --io16 --batch 4 --only "@1280" --iters 20 2>&1 | grep "^{'layer"; C2M_LIB=$lib timeout 300 python scripts/bench_conv.py --io16 --only "64->64 @640" --iters 20 2>&1 | grep "^{'layer"; done) > $O/ab_epi16_layers.log 2>&1
                (for lib in "" $R/build_exp/nomix/libc2m_hip.so "" $R/build_exp/nomix/libc2m_hip.so; do echo "=== C2M_LIB=$lib"; C2M_LIB=$lib timeout 600 python bench.py --lr 320 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-alt 2>&1 | grep "^{" | cut -c1-700; done) > $O/ab_epi16_step.log 2>&1 ;;
    ab_lib)     (for lib in "" $R/build_exp/${AB_LIB:-fastall}/libc2m_hip.so "" $R/build_exp/${AB_LIB:-fastall}/libc2m_hip.so; do echo "=== C2M_LIB=$lib"; C2M_LIB=$lib timeout 300 python scripts/bench_conv.py --algo split16 --iters 20 2>&1 | grep "^{'layer"; done) > $O/ab_lib_layers.log 2>&1
                (for lib in "" $R/build_exp/${AB_LIB:-fastall}/libc2m_hip.so "" $R/build_exp/${AB_LIB:-fastall}/libc2m_hip.so; do echo "=== C2M_LIB=$lib"; C2M_LIB=$lib timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt 2>&1 | grep "^{" | cut -c1-700; done) > $O/ab_lib_step.log 2>&1 ;;
    power3)     (for abl in 0 1 6 8 32 64 39 47 48; do
                   echo "=== time C2M_SPLIT_ABL=$abl"; C2M_SPLIT_ABL=$abl timeout 200 python scripts/bench_conv.py --algo split16 --only "body 64->64 @640" --iters 20 2>&1 | grep "^{'layer"
                   C2M_SPLIT_ABL=$abl bash scripts/power_probe.sh "python $R/scripts/bench_conv.py --algo split16 --iters 5000 --only 'body 64->64 @640'" "power C2M_SPLIT_ABL=$abl (1 no weight DMA, 2 no halo loads, 4 no split, 8 no unit-end waits / barriers, 16 no MFMAs, 32 no operand reads, 64 one store per tile)" 8
                 done) > $O/power_probe3.log 2>&1 ;;
    prof_cfg5)  cd /tmp
                timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_cfg5 -o step -- python $R/bench.py --lr 320 --dtype bf16 --steps 3 --warmup 2 --no-cpu-baseline --no-alt > $O/rocprof_cfg5.log 2>&1
                cp $(find $O/prof_cfg5 -name '*kernel_stats.csv' | head -1) $O/cfg5_kernel_stats.csv; rm -rf $O/prof_cfg5
                cd $R ;;
    pmc)        cd /tmp
                timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace -f csv -d $O/pmc_mfma -o step -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-alt > $O/pmc_mfma.log 2>&1
                timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $O/pmc_fetch -o step -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-alt > $O/pmc_fetch.log 2>&1
                timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $O/pmc_write -o step -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-alt > $O/pmc_write.log 2>&1
                cd $R ;;
    pmc_corr)   cd /tmp
                timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace -f csv -d $O/pmc_corr1 -o c -- python $R/bench.py --workload corr --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_corr1.log 2>&1
                timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace -f csv -d $O/pmc_corr2 -o c -- python $R/bench.py --workload corr --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_corr2.log 2>&1
                timeout 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace -f csv -d $O/pmc_corr3 -o c -- python $R/bench.py --workload corr --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_corr3.log 2>&1
                cd $R
                (python scripts/pmc_kernel.py $O/pmc_corr1 corr; python scripts/pmc_kernel.py $O/pmc_corr2 corr; python scripts/pmc_kernel.py $O/pmc_corr3 corr) > $O/pmc_corr_summary.txt 2>&1
                rm -rf $O/pmc_corr1 $O/pmc_corr2 $O/pmc_corr3 ;;
    pmc_cfg5)   cd /tmp
                timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $O/pmc5_fetch -o s -- python $R/bench.py --lr 320 --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc5_fetch.log 2>&1
                timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $O/pmc5_write -o s -- python $R/bench.py --lr 320 --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc5_write.log 2>&1
                cd $R
                (python scripts/pmc_kernel.py $O/pmc5_fetch c2m; python scripts/pmc_kernel.py $O/pmc5_write c2m) > $O/pmc_cfg5_summary.txt 2>&1
                rm -rf $O/pmc5_fetch $O/pmc5_write ;;
    prof_train) cd /tmp
                timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_train -o t -- python $R/bench.py --workload train --batch 4 --steps 10 --warmup 3 > $O/rocprof_train.log 2>&1
                cd $R
                python - "$O" <<'PY'
import csv, sys, glob
O = sys.argv[1]
f = glob.glob(O + "/prof_train/**/t_kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(O + "/train_b4_kernel_stats.txt", "w") as out:
    out.write("total kernel ms over 13 steps: %.1f, kernels: %d kinds, launches %d\n" % (tot / 1e6, len(rows), sum(int(r["Calls"]) for r in rows)))
    for r in rows[:45]:
        out.write("%-100s calls %6s total_ms %9.3f avg_us %9.2f pct %s\n" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
                rm -rf $O/prof_train ;;
    pmc_train)  cd /tmp
                for c in FETCH_SIZE WRITE_SIZE; do   # (one counter per pass: both together exceed what the hardware collects)
                  timeout 400 rocprofv3 --pmc $c --kernel-trace -f csv -d $O/pmct_$c -o s -- python $R/bench.py --workload train --global-batch 4 --steps 3 --warmup 2 > $O/pmct_$c.log 2>&1
                done
                cd $R
                (python scripts/pmc_kernel.py $O/pmct_FETCH_SIZE ""; python scripts/pmc_kernel.py $O/pmct_WRITE_SIZE "") > $O/pmc_train_summary.txt 2>&1
                rm -rf $O/pmct_FETCH_SIZE $O/pmct_WRITE_SIZE ;;
    abl_cycles) cd /tmp
                for abl in 0 2 32 64 8 111 48 39; do
                  C2M_SPLIT_ABL=$abl timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_LDS --kernel-trace -f csv -d $O/abl_$abl -o c -- python $R/scripts/bench_conv.py --algo split16 --only 'body 64->64 @640' --iters 6 > $O/abl_$abl.log 2>&1
                  echo "=== C2M_SPLIT_ABL=$abl" >> $O/abl_cycles.txt
                  grep "^{'layer" $O/abl_$abl.log >> $O/abl_cycles.txt
                  python $R/scripts/pmc_kernel.py $O/abl_$abl conv3x3_split_kernel >> $O/abl_cycles.txt 2>&1
                  rm -rf $O/abl_$abl
                done
                cd $R ;;
    avail)      cd /tmp; (rocprofv3 --list-avail 2>&1 | grep -o "\b\(TCP\|TA\|TD\|TCC\|SQ\|SQC\|GRBM\|CPC\|SPI\)_[A-Za-z0-9_]*" | sort -u | tr "\n" " ") > $O/pmc_avail.txt 2>&1; cd $R ;;
    abl_rb)     (for hw in 640 320 160; do for m in 0 64 6 2; do
                   echo "=== C2M_SPLIT_ABL=$m hw=$hw (64: one store per tile; 6: no halo loads, no split; 2: no halo loads)"
                   ps=""; [ $m = 0 ] && [ $hw = 640 ] && ps="--persample"
                   C2M_SPLIT_ABL=$m timeout 200 python scripts/abl_resblock.py --hw $hw $ps 2>&1 | grep "^{"
                 done; done) > $O/abl_resblock.log 2>&1 ;;
    pmc_dcn)    cd /tmp
                i=0
                for set in "SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU" \
                           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
                           "${PMC_SET3:-TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum}" \
                           "${PMC_SET4:-TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr}" \
                           "${PMC_SET5:-TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TD_TD_BUSY_sum}" \
                           "${PMC_SET6:-TCP_TCC_WRITE_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_HIT_sum TCC_MISS_sum}"; do
                  i=$((i+1))
                  timeout 300 rocprofv3 --pmc $set --kernel-trace -f csv -d $O/pmcd_$i -o c -- python $R/scripts/bench_dcn_nhwc.py --iters 2 > $O/pmcd_$i.log 2>&1
                  echo "=== dcn forward, pass $i: $set" >> $O/pmc_dcn_c3.txt
                  python $R/scripts/pmc_kernel.py $O/pmcd_$i dcn_fwd >> $O/pmc_dcn_c3.txt 2>&1
                  tail -3 $O/pmcd_$i.log | cut -c1-300 >> $O/pmc_dcn_c3.txt
                  timeout 300 rocprofv3 --pmc $set --kernel-trace -f csv -d $O/pmcc_$i -o c -- python $R/scripts/abl_c3.py 16 640 twin > $O/pmcc_$i.log 2>&1
                  echo "=== first layer (twin), pass $i: $set" >> $O/pmc_dcn_c3.txt
                  python $R/scripts/pmc_kernel.py $O/pmcc_$i conv3x3_c3 >> $O/pmc_dcn_c3.txt 2>&1
                  tail -2 $O/pmcc_$i.log | cut -c1-300 >> $O/pmc_dcn_c3.txt
                  rm -rf $O/pmcd_$i $O/pmcc_$i
                done
                cd $R ;;
    diag_rb)    timeout 600 python scripts/diag_resblock.py ${RB_ARGS:-check time} > $O/diag_resblock.log 2>&1; echo "rc=$?" >> $O/diag_resblock.log ;;
    abl_rbk)    (for m in "" ${RBK_MASKS:-8 16 32 40 56 64 120 2} ""; do echo "=== ${m:-in-tree} (C2M_RB_ABL: 1 no stores, 2 no MFMAs, 4 no x loads, 8 no unit-end waits/barriers, 16 no output epilogue, 32 no operand wait before a unit's first tap, 64 no conv1 epilogue)"; C2M_LIB=${m:+$R/build_exp/libc2m_rb$m.so} timeout 200 python scripts/diag_resblock.py time 2>&1 | grep "^{"; done) > $O/abl_resblock_kernel.log 2>&1 ;;
    abl_rbr)    (for m in 0 ${RBR_MASKS:-1 16 64 80} 0; do echo "=== C2M_RB_ABLR=$m (runtime, no dead-code elimination: 1 no output stores, 16 no output epilogue, 64 no conv1 epilogue)"; C2M_RB_ABLR=$m timeout 200 python scripts/diag_resblock.py time 2>&1 | grep "^{"; done) > $O/abl_resblock_runtime.log 2>&1 ;;
    pmc_rb)     cd /tmp
                i=0
                for set in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" \
                           "SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS" \
                           "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
                  i=$((i+1))
                  timeout 300 rocprofv3 --pmc $set --kernel-trace -f csv -d $O/pmcrb_$i -o c -- python $R/scripts/diag_resblock.py time640 > $O/pmcrb_$i.log 2>&1
                  echo "=== pass $i: $set" >> $O/pmc_resblock.txt
                  grep "^{" $O/pmcrb_$i.log >> $O/pmc_resblock.txt
                  python $R/scripts/pmc_kernel.py $O/pmcrb_$i "conv" >> $O/pmc_resblock.txt 2>&1
                  python - $O/pmcrb_$i <<'PY' >> $O/pmc_resblock.txt 2>&1
import csv, glob, sys, re
csv.field_size_limit(1 << 30)
acc = {}
for fn in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if "conv" not in k: continue
        m = re.search(r"(c2m::[A-Za-z0-9_:]+(<[^>(]*>)?)", k); k = m.group(1) if m else k[:60]
        acc.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in acc.items():
    v.sort(); print("   duration ms under the counters:", k, "median", round(v[len(v)//2], 4), "n", len(v))
PY
                  rm -rf $O/pmcrb_$i
                done
                cd $R ;;
    test_exp)   C2M_LIB=$R/build_exp/exp/libc2m_hip.so timeout 1500 python -m pytest tests/test_conv_gpu.py -m gpu -q -rA -k "wino16 or loader_matrix or fused_residual" 2>&1 | tail -90 > $O/pytest_experimental.log ;;
    ab_corrf)   (for lib in "" cf_slow "" cf_slow; do echo "=== ${lib:-in-tree (third-best threshold in front of the top-3 update)} ${lib:+(C2M_CORRF_FAST=0: unconditional update)}"; C2M_LIB=${lib:+$R/build_exp/$lib/libc2m_hip.so} timeout 120 python scripts/abl_corr_filter.py 2>&1 | grep "^{"; C2M_LIB=${lib:+$R/build_exp/$lib/libc2m_hip.so} timeout 300 python bench.py --workload corr --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep "^{" | python -c "import sys,json; p=json.loads(sys.stdin.read()); print({'configs1_pairs_per_s': round(p['value'],1), 'ms_per_step': round(p['ms_per_step'],3), 'kernels_ms': p['c2m_kernel_ms_per_step']})"; done) > $O/ab_corr_filter_fast.log 2>&1 ;;
    pmc_conv_ta) cd /tmp
                i=0
                for set in "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE" \
                           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum GRBM_GUI_ACTIVE" \
                           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
                           "TD_TD_BUSY_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
                  i=$((i+1))
                  timeout 300 rocprofv3 --pmc $set --kernel-trace -f csv -d $O/pmcta_$i -o c -- python $R/scripts/bench_conv.py --algo split16 --only "64->64 @640" --iters 4 > $O/pmcta_$i.log 2>&1
                  echo "=== pass $i: $set" >> $O/pmc_conv_ta.txt
                  grep "^{'layer" $O/pmcta_$i.log >> $O/pmc_conv_ta.txt
                  python $R/scripts/pmc_kernel.py $O/pmcta_$i "conv3x3_split_kernel" >> $O/pmc_conv_ta.txt 2>&1
                  rm -rf $O/pmcta_$i
                done
                cd $R ;;
    diag_pf1)   C2M_CORR_PF=1 timeout 600 python scripts/diag_corr_filter.py > $O/diag_corr_filter_pf1.log 2>&1 ;;
    *)          echo "unknown stage $stage" ;;
  esac
done
cd $R
find gpurun_out/$TAG -name "*.db" -delete
find gpurun_out/$TAG -name "*kernel_trace.csv" -size +8M -delete
