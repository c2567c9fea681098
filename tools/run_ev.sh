OUT=$GRAFT_REPO_ROOT/gpurun_out/ev
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for f in "--kernels" "" "--kernels" ""; do
  python3 bench.py --no-cpu-baseline --no-entropy $f > $OUT/c2.json 2>/dev/null
  python3 bench.py --no-cpu-baseline $f > $OUT/c3.json 2>/dev/null
  python3 - <<PY
import json
for c in ("c2","c3"):
    d=json.load(open("$OUT/%s.json"%c)); r=d["roofline"]; print("flag='$f'",c,round(d["value"]),round(d["ms_per_step"],3),r["kernel"],r["launches"],round(r["avg_launch_ms"],4),round(r["frac"],4))
PY
done
