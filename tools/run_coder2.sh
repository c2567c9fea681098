set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/coder2
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_entropy.py -m gpu -x -q > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
for rep in 1 2; do
for spw in 4 8 16 2; do
python3 bench.py --no-cpu-baseline --kernels --streams-per-wg $spw > $OUT/c3_$spw.json 2> $OUT/c3_$spw.err
python3 -c "
import json,sys
d=json.loads(open('$OUT/c3_$spw.json').read().strip().splitlines()[-1]); print('spw $spw', round(d['value']), round(d['ms_per_step'],3), d['coder'] and round(d['coder']['ms_per_batch'],2))"; grep "step periods" $OUT/c3_$spw.err
done
python3 bench.py --no-entropy --no-cpu-baseline --kernels > $OUT/c2.json 2> $OUT/c2.err
python3 -c "
import json,sys
d=json.loads(open('$OUT/c2.json').read().strip().splitlines()[-1]); print('c2', round(d['value']), round(d['ms_per_step'],3))"; grep "step periods" $OUT/c2.err
done
