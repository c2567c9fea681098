cd $GRAFT_REPO_ROOT
OUT=gpurun_out/stamps; mkdir -p $OUT
DSIC_EXTRA_FLAGS="-DWB_STAMP=1" python3 domain-specific-image-compression_amd/build.py --force > $OUT/build.log 2>&1 || { tail $OUT/build.log; exit 1; }
DSIC_WINO_FUSED=1 python3 tools/wbx_stamps.py > $OUT/x1.log 2>&1; cat $OUT/x1.log
DSIC_WINO_FUSED=1 LAYER=s2 python3 tools/wbx_stamps.py > $OUT/x2.log 2>&1; tail -26 $OUT/x2.log
python3 domain-specific-image-compression_amd/build.py --force > $OUT/build2.log 2>&1 || { tail $OUT/build2.log; exit 1; }
