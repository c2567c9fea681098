cd $GRAFT_REPO_ROOT
LIB=domain-specific-image-compression_amd/libdsic_hip.so
for V in nopf abl0; do
cp tools/_abl/lib_$V.so $LIB
for i in 1 2 3; do echo -n "$V run $i: "; timeout -k 10 200 python3 -m pytest tests/test_gpu_conv.py -q -k "space_to_depth and bf16" 2>&1 | tail -1; done
done
cp tools/_abl/lib_abl0.so $LIB
timeout -k 10 200 python3 -m pytest tests/test_gpu_conv.py -q -x -k "space_to_depth and bf16 and 192" 2>&1 | grep "Error\|assert\|err" | head
