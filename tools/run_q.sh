cd $GRAFT_REPO_ROOT
OUT=gpurun_out/q; mkdir -p $OUT
python3 bench.py --no-cpu-baseline > $OUT/c3.json 2>/dev/null
python3 - <<PY
import json
d=json.load(open("$OUT/c3.json")); print("plain", round(d["value"]), round(d["ms_per_step"],3))
PY
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/t.json
python3 - <<PY
import json
d=json.load(open("$OUT/t.json")); print("torchrun", round(d["value"]), round(d["ms_per_step"],3))
PY
python3 bench.py --no-cpu-baseline --no-entropy > $OUT/c2.json 2>/dev/null
python3 - <<PY
import json
d=json.load(open("$OUT/c2.json")); print("plain c2", round(d["value"]), round(d["ms_per_step"],3))
PY
