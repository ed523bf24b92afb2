#!/bin/bash
cd "$GRAFT_REPO_ROOT"; o=gpurun_out/r03I; mkdir -p $o
timeout 1500 python tools/fuzz_gpu.py --seeds 25 --start 9000 > $o/fuzz_gpu.log 2>&1; echo "gpu rc=$?"; tail -1 $o/fuzz_gpu.log
timeout 900 python tools/fuzz_sl.py --cases 600 --seed 31 > $o/fuzz_sl.log 2>&1; echo "sl rc=$?"; tail -1 $o/fuzz_sl.log
timeout 900 python tools/fuzz_peaks.py --cases 2000 --seed 31 > $o/fuzz_peaks.log 2>&1; echo "peaks rc=$?"; tail -1 $o/fuzz_peaks.log
