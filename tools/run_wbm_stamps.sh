cd $GRAFT_REPO_ROOT
OUT=gpurun_out/wbm; mkdir -p $OUT
LIB=domain-specific-image-compression_amd/libdsic_hip.so
timeout -k 10 120 python3 tools/wbm_debug.py 2>&1 | grep -v "^tensor\|^  *\[" | tail -3
timeout -k 10 300 python3 -m pytest tests/test_gpu_conv.py -x -q -k "winograd or two_pass or space_to_depth" 2>&1 | tail -5 || exit 1
for L in 3x3 s2 convT; do echo -n "M64=1  "; LAYER=$L REPS=10 python3 tools/wb_layer.py 2>/dev/null | tail -1; done
cp tools/_abl/lib_wbmstamp.so $LIB
for L in 3x3 convT s2; do LAYER=$L timeout -k 10 60 python3 tools/wbm_stamps.py 2>/dev/null | tee $OUT/stamps_$L.txt; done
cp tools/_abl/lib_abl0.so $LIB
