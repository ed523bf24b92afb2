mkdir -p gpurun_out/r03k
{
for lib in libpdc_ab_r02.so libperiodicity_hip.so libpdc_ab_r02.so libperiodicity_hip.so; do echo "$lib"; PDC_LIBRARY=periodicity_amd/$lib python tools/peaks_timing.py; done
} > gpurun_out/r03k/peaks.txt 2>&1
python -m pytest tests/test_peaks_gpu.py -x -q 2>&1 | tail -3 >> gpurun_out/r03k/peaks.txt
cat gpurun_out/r03k/peaks.txt
