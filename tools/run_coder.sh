set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/coder
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_entropy.py tests/test_gpu_fullsize.py -m gpu -x -q > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
python3 tools/entropy_bench.py > $OUT/entropy_bench.log 2>&1 || true
tail -5 $OUT/entropy_bench.log
python3 bench.py --no-cpu-baseline --kernels > $OUT/c3.json 2> $OUT/c3.err
python3 bench.py --no-entropy --no-cpu-baseline --kernels > $OUT/c2.json 2> $OUT/c2.err
python3 bench.py --no-cpu-baseline --size 512 --channels 4 --batch 32 --steps 5 --warmup 2 --kernels > $OUT/c5.json 2> $OUT/c5.err
for f in c2 c3 c5; do python3 -c "
import json,sys
d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value']), round(d['ms_per_step'],3), d['coder'] and round(d['coder']['ms_per_batch'],2), d['coder'] and round(d['coder']['symbols_per_s']/1e6))"; grep "step periods" $OUT/$f.err; done
