cd $GRAFT_REPO_ROOT
G=gpurun_out
python3 bench.py > $G/bench_config3.json 2> $G/bench_config3.err
python3 bench.py --no-entropy > $G/bench_config2.json 2> $G/bench_config2.err
python3 bench.py --size 512 --channels 4 --batch 32 --steps 12 --warmup 3 > $G/bench_config5.json 2> $G/bench_config5.err
DSIC_WINO_BF16=0 python3 bench.py --no-cpu-baseline > $G/bench_config3_fp32kernels.json 2> /dev/null
python3 - <<PY
import json
for c in ("config2","config3","config5","config3_fp32kernels"):
    d=json.load(open("$G/bench_%s.json"%c)); r=d["roofline"]
    print(c, round(d["value"]), round(d["ms_per_step"],3), r["kernel"], round(r["avg_launch_ms"],4), round(r["frac"],4), d.get("cpu_baseline",{}).get("value"))
PY
