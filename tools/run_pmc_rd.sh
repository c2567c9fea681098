cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmcrd; rm -rf $OUT; mkdir -p $OUT
cd $R
timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace -f csv -d $OUT -o rd -- python3 tools/gap_probe.py > $OUT/rd.log 2>&1 || { echo "pass failed rc=$?"; tail -3 $OUT/rd.log; }
ls $OUT
