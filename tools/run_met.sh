OUT=$GRAFT_REPO_ROOT/gpurun_out/met
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for f in "--serial-metrics" "" "--serial-metrics" ""; do
  python3 bench.py --no-cpu-baseline --no-entropy $f > $OUT/c2.json 2>/dev/null
  python3 bench.py --no-cpu-baseline $f > $OUT/c3.json 2>/dev/null
  python3 - <<PY
import json
for c in ("c2","c3"):
    d=json.load(open("$OUT/%s.json"%c)); print("flag='$f'",c,round(d["value"]),d["ms_per_step"],d["mean_bpp"],d["mean_ms_ssim"],d["mean_bpp_coded"])
PY
done
