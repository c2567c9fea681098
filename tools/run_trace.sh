set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace7
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
DSIC_HYPER_STREAM=0 DSIC_WINO_SPLITK=1 rocprofv3 --kernel-trace -f csv -d $OUT -o s1 -- python3 tools/gap_probe.py > $OUT/s1.log 2>&1
DSIC_HYPER_STREAM=1 DSIC_WINO_SPLITK=1 rocprofv3 --kernel-trace -f csv -d $OUT -o f1 -- python3 tools/gap_probe.py > $OUT/f1.log 2>&1
DSIC_HYPER_STREAM=1 DSIC_WINO_SPLITK=0 rocprofv3 --kernel-trace -f csv -d $OUT -o f0 -- python3 tools/gap_probe.py > $OUT/f0.log 2>&1
ls $OUT
