cd $GRAFT_REPO_ROOT
OUT=gpurun_out/stamps; mkdir -p $OUT
DSIC_EXTRA_FLAGS="-DWB_STAMP=1" python3 domain-specific-image-compression_amd/build.py --force > $OUT/build.log 2>&1 || { tail $OUT/build.log; exit 1; }
python3 tools/wb_stamps.py > $OUT/s1.log 2>&1; cat $OUT/s1.log
LAYER=s2 python3 tools/wb_stamps.py > $OUT/s2.log 2>&1; head -16 $OUT/s2.log; tail -3 $OUT/s2.log
python3 domain-specific-image-compression_amd/build.py --force > $OUT/build2.log 2>&1 || { tail $OUT/build2.log; exit 1; }
timeout -k 10 300 python3 -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py -m gpu -q -x > $OUT/pytest.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/pytest.log
for rep in 1 2; do
python3 bench.py --no-entropy --no-cpu-baseline --kernels > $OUT/c2.json 2> $OUT/c2.err
python3 bench.py --no-cpu-baseline --kernels > $OUT/c3.json 2> $OUT/c3.err
for f in c2 c3; do python3 -c "
import json,sys
d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value']), round(d['ms_per_step'],3), 'bpp', d['mean_bpp'], d['mean_ms_ssim'], d['mean_bpp_coded'])"; grep "step periods" $OUT/$f.err; done
done
cat $OUT/c2.err | tail -7
