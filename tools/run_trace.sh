set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace3
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -f csv -d $OUT -o t -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
ls -la $OUT
