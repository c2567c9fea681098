cd $GRAFT_REPO_ROOT
OUT=gpurun_out/wbm; mkdir -p $OUT
LIB=domain-specific-image-compression_amd/libdsic_hip.so
timeout -k 10 300 python3 -m pytest tests/test_gpu_conv.py -x -q -k "winograd or two_pass or space_to_depth" 2>&1 | tail -3
for L in 3x3 s2 convT; do echo -n "new  "; LAYER=$L REPS=10 python3 tools/wb_layer.py 2>/dev/null | tail -1; done
for L in 3x3 s2 convT; do echo -n "old  "; DSIC_WINO_M64=0 LAYER=$L REPS=10 python3 tools/wb_layer.py 2>/dev/null | tail -1; done
cp tools/_abl/lib_wbmstamp.so $LIB
LAYER=3x3 timeout -k 10 60 python3 tools/wbm_stamps.py 2>/dev/null | tee $OUT/stamps_3x3.txt
cp tools/_abl/lib_abl0.so $LIB
