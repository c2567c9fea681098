cd $GRAFT_REPO_ROOT
python3 tools/entropy_bench.py 2>/dev/null | grep -E "compress|tables"
python3 -m pytest tests/test_gpu_entropy.py -m gpu -q -x 2>&1 | tail -1
