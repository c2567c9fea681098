# timing of the ablation builds of convT_image.hip (tools/_abl/lib_img<N>.so = -DIMG_ABL=<N>, built in the container)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/img_abl; mkdir -p $OUT
LIB=domain-specific-image-compression_amd/libdsic_hip.so
cp $LIB /tmp/lib_keep.so
for A in $ABLS; do
  cp tools/_abl/lib_img$A.so $LIB
  echo -n "IMG_ABL=$A  " | tee -a $OUT/abl.log
  LAYER=image REPS=20 python3 tools/wb_layer.py 2>&1 | tail -1 | tee -a $OUT/abl.log
done
cp /tmp/lib_keep.so $LIB
