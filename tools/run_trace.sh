# kernel trace of a few forward passes (no HIP events, no coder): per-launch timeline of the main stream
set -e
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace; rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -f csv -d $OUT -o t -- python3 tools/gap_probe.py > $OUT/t.log 2>&1
python3 tools/step_timeline.py $(find $OUT -name "*kernel_trace.csv" | head -1) | tee $OUT/timeline.txt
