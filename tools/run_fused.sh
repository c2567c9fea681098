cd $GRAFT_REPO_ROOT
OUT=gpurun_out/fused; mkdir -p $OUT
for F in "-DWBX_RING=2 -DWBX_ADOUBLE=0" "-DWBX_RING=4 -DWBX_ADOUBLE=0"; do
DSIC_EXTRA_FLAGS="$F" python3 domain-specific-image-compression_amd/build.py --force > $OUT/build.log 2>&1 || { tail $OUT/build.log; exit 1; }
DSIC_WINO_FUSED=1 timeout -k 10 300 python3 tools/det_probe.py 2>&1 | grep -E "bad channels" | cut -c1-100
for L in 3x3 s2; do
echo -n "$F FUSED=1 "
DSIC_WINO_FUSED=1 LAYER=$L timeout -k 10 120 python3 tools/wb_layer.py 2>/dev/null | tail -1
done
done
python3 domain-specific-image-compression_amd/build.py --force > $OUT/build2.log 2>&1
