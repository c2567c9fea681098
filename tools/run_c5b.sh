cd $GRAFT_REPO_ROOT
OUT=gpurun_out/c5; mkdir -p $OUT
timeout -k 10 300 python3 -m pytest tests/test_gpu_conv.py -m gpu -q -x -k direct_split > $OUT/pytest2.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/pytest2.log
for rep in 1 2; do
for D in 1 0; do
DSIC_DIRECT_5S2=$D python3 bench.py --no-entropy --no-cpu-baseline --kernels > $OUT/c2_$D.json 2> $OUT/c2_$D.err
python3 -c "
import json
d=json.loads(open('$OUT/c2_$D.json').read().strip().splitlines()[-1]); print('direct=$D c2', round(d['value']), round(d['ms_per_step'],3), d['mean_bpp'], d['mean_ms_ssim'])"
done
done
tail -8 $OUT/c2_1.err
