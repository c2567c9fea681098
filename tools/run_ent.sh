cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ent; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_entropy.py -m gpu -q -x > $OUT/pytest.log 2>&1
echo "tests rc=$?"; tail -4 $OUT/pytest.log
for i in 1 2; do
python3 bench.py --no-cpu-baseline > $OUT/c3.json 2>/dev/null
python3 - <<PY
import json
d=json.load(open("$OUT/c3.json")); print("c3", round(d["value"]), round(d["ms_per_step"],3), round(d["coder"]["ms_per_batch"],3), d["mean_bpp_coded"])
PY
done
