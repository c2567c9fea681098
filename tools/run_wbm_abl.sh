cd $GRAFT_REPO_ROOT
OUT=gpurun_out/wbm; mkdir -p $OUT
LIB=domain-specific-image-compression_amd/libdsic_hip.so
for A in 1 2; do
cp tools/_abl/lib_wbmabl$A.so $LIB
echo "=== WBM_ABL=$A"
LAYER=3x3 timeout -k 10 60 python3 tools/wbm_stamps.py 2>/dev/null | tee $OUT/abl${A}_3x3.txt
done
cp tools/_abl/lib_abl0.so $LIB
