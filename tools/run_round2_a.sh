set -e
mkdir -p gpurun_out/r2a
python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1 || { tail -40 gpurun_out/r2a/pytest.log; exit 1; }
tail -3 gpurun_out/r2a/pytest.log
python bench.py --kernels > gpurun_out/r2a/bench3.json 2> gpurun_out/r2a/bench3.err
tail -c 3000 gpurun_out/r2a/bench3.json
python bench.py --no-entropy --no-cpu-baseline > gpurun_out/r2a/bench2.json 2> gpurun_out/r2a/bench2.err
python bench.py --size 512 --channels 4 --batch 32 --steps 5 --warmup 2 --kernels > gpurun_out/r2a/bench5.json 2> gpurun_out/r2a/bench5.err
tail -c 1500 gpurun_out/r2a/bench5.json
