mkdir -p gpurun_out/r03l
{
for lib in libperiodicity_hip.so libpdc_ab_pk6.so libpdc_ab_pk8.so libperiodicity_hip.so libpdc_ab_pk6.so; do echo "$lib"; PDC_LIBRARY=periodicity_amd/$lib python tools/peaks_timing.py; done
} > gpurun_out/r03l/peaks.txt 2>&1
cat gpurun_out/r03l/peaks.txt
