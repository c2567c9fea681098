# A/B of library builds placed in tools/_abl/ (built in the container) on one box:
#   LIBS="base pksub" bash tools/run_ab_libs.sh     -> one line per build and layer type, then a config-2 bench line per build
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_libs; mkdir -p $OUT
LIB=domain-specific-image-compression_amd/libdsic_hip.so
cp $LIB /tmp/lib_keep.so
for A in $LIBS $LIBS; do
  cp tools/_abl/lib_$A.so $LIB
  for L in 3x3 s2 convT; do
    echo -n "$A  " | tee -a $OUT/ab.log
    LAYER=$L REPS=10 python3 tools/wb_layer.py 2>/dev/null | tail -1 | tee -a $OUT/ab.log
  done
done
for A in $LIBS; do
  cp tools/_abl/lib_$A.so $LIB
  echo -n "$A  " | tee -a $OUT/ab.log
  python3 bench.py --no-entropy --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 2', round(d['value']), 'patches/s', round(d['ms_per_step'],3), 'ms')" | tee -a $OUT/ab.log
done
cp /tmp/lib_keep.so $LIB
