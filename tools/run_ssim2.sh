cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ssim; mkdir -p $OUT
for RB in 8 12 16; do
DSIC_SSIM_RB=$RB rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof$RB -o s -- python3 tools/ssim_bench.py > $OUT/prof$RB.log 2>&1
echo "== RB=$RB"; tail -1 $OUT/prof$RB.log
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/prof$RB/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
done
