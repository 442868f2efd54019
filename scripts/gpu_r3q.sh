set -x
mkdir -p gpurun_out/r3q
O=gpurun_out/r3q
for a in 0 1 2 4 6; do
  echo "ABL=$a" >> $O/abl.log
  C2M_SPLIT_ABL=$a timeout 60 python scripts/bench_conv.py --algo split --iters 5 --only "body 64->64 @640" 2>&1 | grep "^{'layer" >> $O/abl.log
done
cat $O/abl.log
R=$GRAFT_REPO_ROOT
cd /tmp
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM"; do
  n=$(echo $pass | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $pass --kernel-trace -f csv -d $R/$O/pmc_$n -o x -- python $R/scripts/bench_conv.py --algo split --iters 3 --only "body 64->64 @640" > $R/$O/pmc_$n.log 2>&1
done
cd $R
python scripts/pmc_kernel.py $O conv3x3_split > $O/summary.txt 2>&1
cat $O/summary.txt
find $O -name "*.db" -delete
