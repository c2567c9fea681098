cd $GRAFT_REPO_ROOT
OUT=gpurun_out/c5; mkdir -p $OUT
timeout -k 10 200 python3 -m pytest tests/test_gpu_conv.py -m gpu -q -x -s -k "direct_split" > $OUT/pytest.log 2>&1
echo "tests rc=$?"; grep -E "direct bf16|passed|failed|Error|error" $OUT/pytest.log | tail -12
