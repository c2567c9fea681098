# in-kernel cycle stamps of conv_wino_bf16m.hip (tools/_abl/lib_wbmstamp.so = a -DWBM_STAMP=1 build made in the container)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/wbm; mkdir -p $OUT
LIB=domain-specific-image-compression_amd/libdsic_hip.so
cp $LIB /tmp/lib_keep.so
cp tools/_abl/lib_wbmstamp.so $LIB
for L in 3x3 convT s2; do LAYER=$L timeout -k 10 60 python3 tools/wbm_stamps.py 2>/dev/null | tee $OUT/stamps_$L.txt; done
cp /tmp/lib_keep.so $LIB
