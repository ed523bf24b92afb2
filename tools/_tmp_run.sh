mkdir -p gpurun_out/r03ae
timeout 1500 python tools/fuzz_gpu.py --seeds 60 --start 5000 > gpurun_out/r03ae/fuzz_gpu.log 2>&1; echo "gpu rc=$?" >> gpurun_out/r03ae/summary.txt
timeout 600 python tools/fuzz_peaks.py --cases 1000 --seed 77 > gpurun_out/r03ae/fuzz_peaks.log 2>&1; echo "peaks rc=$?" >> gpurun_out/r03ae/summary.txt
timeout 900 python tools/fuzz_sl.py --cases 300 --seed 78 > gpurun_out/r03ae/fuzz_sl.log 2>&1; echo "sl rc=$?" >> gpurun_out/r03ae/summary.txt
cat gpurun_out/r03ae/summary.txt; tail -2 gpurun_out/r03ae/fuzz_gpu.log; tail -1 gpurun_out/r03ae/fuzz_peaks.log; tail -1 gpurun_out/r03ae/fuzz_sl.log
