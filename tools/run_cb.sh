OUT=$GRAFT_REPO_ROOT/gpurun_out/cb
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py tests/test_gpu_fullsize.py -m gpu -q -x > $OUT/pytest.log 2>&1
echo "tests rc=$?"; tail -15 $OUT/pytest.log
for f in 1 2; do
  python3 bench.py --no-cpu-baseline --no-entropy > $OUT/c2.json 2>/dev/null
  python3 bench.py --no-cpu-baseline > $OUT/c3.json 2>/dev/null
  python3 - <<PY
import json
for c in ("c2","c3"):
    d=json.load(open("$OUT/%s.json"%c)); r=d["roofline"]; print(c,round(d["value"]),d["ms_per_step"],d["mean_bpp"], r["avg_launch_ms"], r["frac"])
PY
done
