#!/bin/bash
# Profiles of the default bench command (BASELINE config 3) on the GPU box; run through gpurun:
#   gpurun -- 'bash tools/profile_round.sh'
# then, in the build container:  python tools/pmc_summary.py round3 <commit>
# Passes (each its own run; counters never share a run with a trace domain other than --kernel-trace):
#   p1  rocprofv3 --kernel-trace --stats            -> per-kernel durations (profiles/<tag>_kernel_stats.csv)
#   p2  rocprofv3 --pmc FETCH_SIZE --kernel-trace   -> bytes fetched per launch (KiB; x2 on gfx950, see MI355X_MICROARCH.md)
#   p3  rocprofv3 --pmc WRITE_SIZE --kernel-trace   -> bytes written per launch (KiB)
#   p5  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace  -> MFMA-busy fraction
cd /tmp && export TMPDIR=/tmp
# read by the HIP runtime when it initialises - under rocprofv3 that is before python starts (the profiler's preloaded
# library touches the GPU first), so bench.py's own setdefault would come too late
export GPU_MAX_HW_QUEUES=8
R=$GRAFT_REPO_ROOT; G=$R/gpurun_out
cd $R
rm -rf $G/p1 $G/p2 $G/p3 $G/p5
BENCH="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline"
echo "p1"; timeout -k 10 240 rocprofv3 --kernel-trace --stats -f csv -d $G/p1 -o bench -- $BENCH > $G/p1_bench.log 2>&1 || tail -3 $G/p1_bench.log
echo "p2"; timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $G/p2 -o fetch -- $BENCH > $G/p2_bench.log 2>&1 || tail -3 $G/p2_bench.log
echo "p3"; timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $G/p3 -o write -- $BENCH > $G/p3_bench.log 2>&1 || tail -3 $G/p3_bench.log
echo "p5"; timeout -k 10 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -f csv -d $G/p5 -o mfma -- $BENCH > $G/p5_bench.log 2>&1 || tail -3 $G/p5_bench.log
echo "plain runs"
python3 bench.py --decode --e2e > $G/bench_config3.json 2> $G/bench_config3.err
python3 bench.py --no-entropy --no-cpu-baseline > $G/bench_config2.json 2> $G/bench_config2.err
python3 bench.py --size 512 --channels 4 --batch 32 --steps 12 --warmup 3 --no-cpu-baseline > $G/bench_config5.json 2> $G/bench_config5.err
python3 bench.py --spatial-params --no-cpu-baseline > $G/bench_spatial.json 2> $G/bench_spatial.err
python3 bench.py --no-stagger --no-cpu-baseline > $G/bench_config3_nostagger.json 2> /dev/null
python3 bench.py --steps 30 --no-cpu-baseline > $G/bench_config3_30steps.json 2> /dev/null
ls $G/p1 $G/p2 $G/p3 $G/p5
tail -c 600 $G/bench_config5.json
python3 tools/fixture_parity.py > $G/parity_bf16.txt 2>/dev/null
DSIC_WINO_BF16=0 python3 tools/fixture_parity.py > $G/parity_fp32.txt 2>/dev/null
DSIC_WINO_BF16=0 python3 bench.py --no-cpu-baseline > $G/bench_config3_fp32kernels.json 2> /dev/null
DSIC_WINO_M64=0 python3 bench.py --no-cpu-baseline > $G/bench_config3_round2_winograd.json 2> /dev/null
python3 -m pytest tests/test_gpu_bench_parity.py -q -s 2>&1 | grep "histogram\|z flip\|passed\|failed" > $G/parity_bench_batches.txt
