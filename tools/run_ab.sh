set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/ab
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for i in 1 2; do
python3 bench.py --no-entropy --no-cpu-baseline --kernels > $OUT/c2_$i.json 2> $OUT/c2_$i.err
python3 bench.py --no-cpu-baseline --kernels > $OUT/c3_$i.json 2> $OUT/c3_$i.err
done
for f in c2_1 c3_1 c2_2 c3_2; do python3 -c "
import json,sys
d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value']), round(d['ms_per_step'],3), d['coder'] and round(d['coder']['ms_per_batch'],2))"; grep "step periods" $OUT/$f.err; done
