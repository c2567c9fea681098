cd $GRAFT_REPO_ROOT
OUT=gpurun_out/img; mkdir -p $OUT
timeout -k 10 300 python3 -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py tests/test_gpu_fullsize.py -m gpu -q -x > $OUT/pytest.log 2>&1
echo "tests rc=$?"; tail -6 $OUT/pytest.log
for rep in 1 2; do
python3 bench.py --no-entropy --no-cpu-baseline --kernels > $OUT/c2.json 2> $OUT/c2.err
python3 -c "
import json
d=json.loads(open('$OUT/c2.json').read().strip().splitlines()[-1]); print('c2', round(d['value']), round(d['ms_per_step'],3), d['mean_bpp'], d['mean_ms_ssim'])"
done
tail -8 $OUT/c2.err
