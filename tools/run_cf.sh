cd $GRAFT_REPO_ROOT
OUT=gpurun_out/cf; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_abi.py -m gpu -q -x > $OUT/pytest.log 2>&1
echo "tests rc=$?"; tail -6 $OUT/pytest.log
python3 tools/fixture_parity.py 2>/dev/null | tail -11
for i in 1 2; do
python3 bench.py --no-cpu-baseline --no-entropy --kernels > $OUT/c2.json 2> $OUT/c2.err
python3 bench.py --no-cpu-baseline > $OUT/c3.json 2>/dev/null
grep conv_first $OUT/c2.err
python3 - <<PY
import json
for c in ("c2","c3"):
    d=json.load(open("$OUT/%s.json"%c)); print(c, round(d["value"]), round(d["ms_per_step"],3), d["mean_bpp"], d["mean_ms_ssim"])
PY
done
