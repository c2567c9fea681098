OUT=$GRAFT_REPO_ROOT/gpurun_out/full
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1
echo "tests rc=$?"; tail -8 $OUT/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
python3 bench.py --kernels > $OUT/c3.json 2> $OUT/c3.err
tail -c 2500 $OUT/c3.json
