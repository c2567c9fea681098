OUT=$GRAFT_REPO_ROOT/gpurun_out/sk
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py tests/test_gpu_fullsize.py -m gpu -q -x > $OUT/pytest.log 2>&1
echo "tests rc=$?"; tail -15 $OUT/pytest.log
for f in 0 1 0 1; do
  DSIC_WINO_SPLITK=$f python3 bench.py --no-cpu-baseline --no-entropy > $OUT/c2_$f.json 2>/dev/null
  DSIC_WINO_SPLITK=$f python3 bench.py --no-cpu-baseline > $OUT/c3_$f.json 2>/dev/null
  python3 - <<PY
import json
for c in ("c2","c3"):
    d=json.load(open("$OUT/%s_$f.json"%c)); print("splitk=$f",c,round(d["value"]),d["ms_per_step"],d["mean_bpp"])
PY
done
