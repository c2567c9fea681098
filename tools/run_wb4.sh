OUT=$GRAFT_REPO_ROOT/gpurun_out/wb4
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py -m gpu -q -x > $OUT/pytest.log 2>&1
echo "tests rc=$?"; tail -5 $OUT/pytest.log
for L in 3x3 s2 convT; do echo -n "LAYER=$L  "; LAYER=$L python3 tools/wb_layer.py 2>/dev/null | tail -1; done
for rep in 1 2; do
python3 bench.py --no-entropy --no-cpu-baseline --kernels > $OUT/c2.json 2> $OUT/c2.err
python3 bench.py --no-cpu-baseline --kernels > $OUT/c3.json 2> $OUT/c3.err
for f in c2 c3; do python3 -c "
import json,sys
d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value']), round(d['ms_per_step'],3), 'bpp', d['mean_bpp'], d['mean_ms_ssim'], d['mean_bpp_coded'])"; grep "step periods" $OUT/$f.err; done
done
cat $OUT/c2.err | tail -7
