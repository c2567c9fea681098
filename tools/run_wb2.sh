OUT=$GRAFT_REPO_ROOT/gpurun_out/wb2
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1
echo "tests rc=$?"; tail -15 $OUT/pytest.log
for rep in 1 2; do
for v in 1 0; do
DSIC_WINO_BF16=$v python3 bench.py --no-entropy --no-cpu-baseline --kernels > $OUT/c2_bf$v.json 2> $OUT/c2_bf$v.err
DSIC_WINO_BF16=$v python3 bench.py --no-cpu-baseline --kernels > $OUT/c3_bf$v.json 2> $OUT/c3_bf$v.err
for f in c2_bf$v c3_bf$v; do python3 -c "
import json,sys
d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value']), round(d['ms_per_step'],3), 'bpp', d['mean_bpp'], d['mean_ms_ssim'], d['mean_bpp_coded'])"; grep "step periods" $OUT/$f.err; done
done
done
cat $OUT/c2_bf1.err | tail -8
