cd $GRAFT_REPO_ROOT
OUT=gpurun_out/knob; mkdir -p $OUT
for s in 2 4 8 16 4 8; do
  python3 bench.py --no-cpu-baseline --streams-per-wg $s > $OUT/k.json 2>/dev/null
  python3 - <<PY
import json
d=json.load(open("$OUT/k.json")); print("streams_per_wg=$s", round(d["value"]), round(d["ms_per_step"],3), round(d["coder"]["ms_per_batch"],3))
PY
done
