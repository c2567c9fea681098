cd $GRAFT_REPO_ROOT
OUT=gpurun_out/img; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py -m gpu -q -x -k "image or fixture or oracle_live" > $OUT/pytest.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/pytest.log
for i in 1 2; do
python3 bench.py --no-cpu-baseline --no-entropy --kernels > $OUT/c2.json 2> $OUT/c2.err
grep convT_image $OUT/c2.err
python3 - <<PY
import json
d=json.load(open("$OUT/c2.json")); print("c2", round(d["value"]), round(d["ms_per_step"],3), d["mean_bpp"])
PY
done
