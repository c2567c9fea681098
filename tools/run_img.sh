cd $GRAFT_REPO_ROOT
OUT=gpurun_out/img; mkdir -p $OUT
timeout -k 10 300 python3 -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py tests/test_gpu_metrics.py -m gpu -q -x -s > $OUT/pytest.log 2>&1
echo "tests rc=$?"; grep -E "image layer|passed|failed|Error" $OUT/pytest.log | tail -12
for rep in 1 2; do
python3 bench.py --no-entropy --no-cpu-baseline --kernels > $OUT/c2.json 2> $OUT/c2.err
python3 -c "
import json
d=json.loads(open('$OUT/c2.json').read().strip().splitlines()[-1]); print('c2', round(d['value']), round(d['ms_per_step'],3), d['mean_bpp'], d['mean_ms_ssim'])"
done
grep -E "convT_image|conv_first" $OUT/c2.err
