cd $GRAFT_REPO_ROOT
OUT=gpurun_out/m64; mkdir -p $OUT
for M in 0 1; do
  for L in 3x3 s2 convT; do
    echo -n "M64=$M  " | tee -a $OUT/layers.log
    DSIC_WINO_M64=$M LAYER=$L REPS=10 python3 tools/wb_layer.py 2>/dev/null | tail -1 | tee -a $OUT/layers.log
  done
  echo -n "M64=$M HW=64 " | tee -a $OUT/layers.log; DSIC_WINO_M64=$M LAYER=3x3 HW=64 REPS=10 python3 tools/wb_layer.py 2>/dev/null | tail -1 | tee -a $OUT/layers.log
  echo -n "M64=$M HW=32 (mode 2: 16 items) " | tee -a $OUT/layers.log; DSIC_WINO_M64=$((M*2)) LAYER=3x3 HW=32 REPS=10 python3 tools/wb_layer.py 2>/dev/null | tail -1 | tee -a $OUT/layers.log
done
DSIC_WINO_M64=0 python3 bench.py --no-cpu-baseline > $OUT/bench_m0.json 2>$OUT/bench.err; python3 -c "import json;d=json.load(open('$OUT/bench_m0.json'));print('M64=0', d['value'], d['ms_per_step'])"
python3 bench.py --no-cpu-baseline > $OUT/bench_m1.json 2>$OUT/bench.err; python3 -c "import json;d=json.load(open('$OUT/bench_m1.json'));print('M64=1', d['value'], d['ms_per_step'], d['mean_bpp'], d['mean_ms_ssim'])"
