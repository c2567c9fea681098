cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ssim; mkdir -p $OUT
python3 -m pytest tests/test_gpu_metrics.py -x -q 2>&1 | tail -5 | tee $OUT/pytest.log
for RB in 0 16 24 32 42 50 64; do echo -n "RB=$RB "; DSIC_SSIM_RB=$RB python3 tools/ssim_bench.py 2>/dev/null | tail -1; done | tee $OUT/rb.log
B=32 C=4 HW=512 python3 tools/ssim_bench.py 2>/dev/null | tail -1 | tee -a $OUT/rb.log
