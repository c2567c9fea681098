cd $GRAFT_REPO_ROOT
OUT=gpurun_out/c5; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_fullsize.py -m gpu -q -x > $OUT/pytest.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/pytest.log
for i in 1 2; do
python3 bench.py --size 512 --channels 4 --batch 32 --steps 12 --warmup 3 --no-cpu-baseline > $OUT/c5.json 2>/dev/null
python3 - <<PY
import json
d=json.load(open("$OUT/c5.json")); r=d["roofline"]; print("c5", round(d["value"]), round(d["ms_per_step"],3), r["kernel"], round(r["frac"],4), d["coder"]["ms_per_batch"])
PY
done
