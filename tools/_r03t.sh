mkdir -p gpurun_out/r03t
{
for lib in libperiodicity_hip.so libpdc_ab_sl_zero.so libpdc_ab_sl_unroll2.so libpdc_ab_sl_both2.so libpdc_ab_sl_both3.so libperiodicity_hip.so libpdc_ab_sl_zero.so; do echo "$lib"; PDC_LIBRARY=periodicity_amd/$lib SHAPES="50000x100000,25000x100000" python tools/sl_shapes.py; done
} > gpurun_out/r03t/sl.txt 2>&1
cat gpurun_out/r03t/sl.txt
