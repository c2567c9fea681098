cd $GRAFT_REPO_ROOT
p() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d.get('coder') or {}; print('$1', round(d['value']), round(d['ms_per_step'],3), c.get('ms_per_batch'))"; }
python3 bench.py --no-cpu-baseline --streams-per-wg 1 | p spw1
python3 bench.py --no-cpu-baseline --streams-per-wg 4 | p spw4
python3 bench.py --no-cpu-baseline --streams-per-wg 1 --steps 30 | p spw1_30
python3 bench.py --no-cpu-baseline --streams-per-wg 1 --coder-cus 64 | p spw1_cus64
python3 bench.py --no-cpu-baseline --streams-per-wg 1 --coder-cus 128 | p spw1_cus128
python3 bench.py --no-cpu-baseline --streams-per-wg 1 --coder-depth 3 | p spw1_depth3
python3 bench.py --no-cpu-baseline --streams-per-wg 1 --size 512 --channels 4 --batch 32 --steps 12 | p c5_spw1
python3 bench.py --no-cpu-baseline --streams-per-wg 4 --size 512 --channels 4 --batch 32 --steps 12 | p c5_spw4
