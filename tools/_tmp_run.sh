cd "$GRAFT_REPO_ROOT"; o=gpurun_out/r03G; mkdir -p $o
timeout 900 python -m pytest tests/test_random_gpu.py -x -q -m gpu -k binned 2>&1 | tail -15
timeout 2400 python tools/fuzz_gpu.py --seeds 12 --start 5000 > $o/fuzz_gpu.log 2>&1; echo "fuzz rc=$?"; tail -3 $o/fuzz_gpu.log
