cd $GRAFT_REPO_ROOT
OUT=gpurun_out/abl; mkdir -p $OUT
for L in 3x3 s2 convT; do
for D in 0 1 2 4 6 8 16 17 31; do
echo -n "LAYER=$L DBG=$D  "
LAYER=$L DSIC_WB_DBG=$D python3 tools/wb_layer.py 2>/dev/null | tail -1
done
done
