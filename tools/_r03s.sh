mkdir -p gpurun_out/r03s
timeout 900 python tools/fuzz_peaks.py --cases 1500 --seed 31 > gpurun_out/r03s/fuzz_peaks.log 2>&1; echo "peaks rc=$?" >> gpurun_out/r03s/summary.txt
timeout 1200 python tools/fuzz_sl.py --cases 400 --seed 32 > gpurun_out/r03s/fuzz_sl.log 2>&1; echo "sl rc=$?" >> gpurun_out/r03s/summary.txt
timeout 1200 python tools/fuzz_gpu.py --seeds 30 --start 3000 > gpurun_out/r03s/fuzz_gpu.log 2>&1; echo "gpu rc=$?" >> gpurun_out/r03s/summary.txt
cat gpurun_out/r03s/summary.txt; tail -3 gpurun_out/r03s/fuzz_peaks.log; tail -3 gpurun_out/r03s/fuzz_sl.log; tail -3 gpurun_out/r03s/fuzz_gpu.log
