cd $GRAFT_REPO_ROOT
OUT=gpurun_out/lanes; mkdir -p $OUT
for CFG in "1 2 8" "2 2 8" "2 3 12" "2 4 12" "2 4 16"; do
set -- $CFG
GPU_MAX_HW_QUEUES=$3 python3 bench.py --no-cpu-baseline --lanes $1 --coder-depth $2 > $OUT/c3.json 2>/dev/null
python3 - <<PY
import json
d=json.load(open("$OUT/c3.json")); print("lanes=$1 depth=$2 queues=$3 c3", round(d["value"]), round(d["ms_per_step"],3), round(d["coder"]["ms_per_batch"],2))
PY
done
