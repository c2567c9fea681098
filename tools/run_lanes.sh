cd $GRAFT_REPO_ROOT
OUT=gpurun_out/lanes; mkdir -p $OUT
for L in 1 2; do
python3 bench.py --no-cpu-baseline --lanes $L --steps 40 --kernels > $OUT/c3.json 2> $OUT/c3.err
python3 - <<PY
import json
d=json.load(open("$OUT/c3.json")); print("lanes=$L steps=40 c3", round(d["value"]), round(d["ms_per_step"],3))
PY
head -2 $OUT/c3.err | tail -1 | cut -c1-200
done
