cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ssim; mkdir -p $OUT
python3 -m pytest tests/test_gpu_metrics.py -x -q 2>&1 | tail -3
prof() {
rocprofv3 --kernel-trace --stats -f csv -d $OUT/p_$1 -o s -- python3 tools/ssim_bench.py > $OUT/p_$1.log 2>&1
tail -1 $OUT/p_$1.log
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/p_$1/**/*kernel_stats.csv",recursive=True)[0]
print("$1", " ".join(f'{r["Name"].split("<")[1][:3]}:{float(r["AverageNs"])/1000:.1f}us' for r in csv.DictReader(open(f)) if "ssim_level" in r["Name"]))
PY
}
prof final
python3 tools/ssim_bench.py | tail -1
B=32 C=4 HW=512 python3 tools/ssim_bench.py | tail -1
