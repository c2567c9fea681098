cd $GRAFT_REPO_ROOT
for i in 1 2; do for D in 1 0; do echo -n "DSIC_IMG_DMA=$D "; DSIC_IMG_DMA=$D python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3))"; done; done
