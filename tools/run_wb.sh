OUT=$GRAFT_REPO_ROOT/gpurun_out/wb
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 -m pytest tests/test_gpu_conv.py -m gpu -q -s -k "winograd or Winograd or space_to_depth" > $OUT/pytest_conv.log 2>&1
echo "conv tests rc=$?"; grep -E "passed|failed|err/" $OUT/pytest_conv.log | tail -70
