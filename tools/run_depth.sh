cd $GRAFT_REPO_ROOT
OUT=gpurun_out/depth; mkdir -p $OUT
for D in 1 2; do
python3 bench.py --no-cpu-baseline --kernels --coder-depth $D --size 512 --channels 4 --batch 32 --steps 12 --warmup 3 > $OUT/c5_$D.json 2> $OUT/c5_$D.err
for f in c5_$D; do python3 -c "
import json
d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value']), round(d['ms_per_step'],3), d['mean_bpp_coded'], round(d['coder']['ms_per_batch'],2))"; grep "step periods" $OUT/$f.err; done
done
python3 bench.py --no-cpu-baseline --kernels --no-entropy --size 512 --channels 4 --batch 32 --steps 12 --warmup 3 2>&1 >/dev/null | grep "step periods"
