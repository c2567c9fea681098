# phase stamps of convT_image_dma_kernel; tools/_abl/lib_imgstamp.so is built in the container with -DIMG_STAMP=1
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/img_stamps; mkdir -p $OUT
LIB=domain-specific-image-compression_amd/libdsic_hip.so
cp $LIB /tmp/lib_keep.so
cp tools/_abl/lib_imgstamp.so $LIB
python3 tools/img_stamps.py 2>/dev/null | tee $OUT/stamps.txt
cp /tmp/lib_keep.so $LIB
