cd $GRAFT_REPO_ROOT
OUT=gpurun_out/knob; mkdir -p $OUT
for F in "-DWB_HPRIO=0" "-DWB_HPRIO=3"; do
DSIC_EXTRA_FLAGS="$F" python3 domain-specific-image-compression_amd/build.py --force > $OUT/build.log 2>&1 || { tail $OUT/build.log; exit 1; }
for L in 3x3 s2 convT; do echo -n "$F LAYER=$L  "; LAYER=$L python3 tools/wb_layer.py 2>/dev/null | tail -1; done
python3 bench.py --no-entropy --no-cpu-baseline > $OUT/c2.json 2>/dev/null; python3 -c "
import json
d=json.loads(open('$OUT/c2.json').read().strip().splitlines()[-1]); print('c2', round(d['value']), round(d['ms_per_step'],3))"
done
python3 domain-specific-image-compression_amd/build.py --force > $OUT/build2.log 2>&1
