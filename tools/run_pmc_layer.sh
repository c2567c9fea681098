cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmcl; mkdir -p $OUT
cd $R
for L in 3x3 s2; do
export LAYER=$L
i=2
for C in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
i=$((i+1))
echo "pass $L $i: $C"
timeout -k 10 100 rocprofv3 --pmc $C --kernel-trace -f csv -d $OUT -o ${L}_p$i -- python3 tools/wb_layer.py > $OUT/${L}_p$i.log 2>&1 || { echo "pass failed rc=$?"; tail -2 $OUT/${L}_p$i.log; }
done
done
ls $OUT | grep counter
