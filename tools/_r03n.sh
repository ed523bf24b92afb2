mkdir -p gpurun_out/r03n
{
for lib in libperiodicity_hip.so libpdc_ab_sh48.so libpdc_ab_sh64.so libperiodicity_hip.so libpdc_ab_sh64.so; do echo "$lib"; PDC_LIBRARY=periodicity_amd/$lib python tools/bench_configs.py --reps 3 --only c3shared,c3sharednoerr | cut -c1-400; done
} > gpurun_out/r03n/shared.txt 2>&1
cat gpurun_out/r03n/shared.txt
