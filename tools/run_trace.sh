set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace4
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -f csv -d $OUT -o c2 -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-entropy > $OUT/c2.json 2> $OUT/c2.err
rocprofv3 --kernel-trace --stats -f csv -d $OUT -o c3 -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/c3.json 2> $OUT/c3.err
ls $OUT
