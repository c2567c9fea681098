# A/B of compile-time ablation builds (tools/_abl/lib_abl<N>.so, built in the container) on one box.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/abl2; mkdir -p $OUT
LIB=domain-specific-image-compression_amd/libdsic_hip.so
cp $LIB /tmp/lib_keep.so
for A in 0 64 128 0; do
  cp tools/_abl/lib_abl$A.so $LIB
  for L in 3x3 s2 convT; do
    echo -n "ABL=$A  " | tee -a $OUT/abl.log
    LAYER=$L REPS=10 python3 tools/wb_layer.py 2>/dev/null | tail -1 | tee -a $OUT/abl.log
  done
done
cp /tmp/lib_keep.so $LIB
python3 bench.py --no-cpu-baseline > $OUT/bench.json 2>$OUT/bench.err; cat $OUT/bench.json
