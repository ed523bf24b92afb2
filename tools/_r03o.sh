mkdir -p gpurun_out/r03o
{
for lib in libperiodicity_hip.so libpdc_ab_slg8_4.so libpdc_ab_slg4_8.so libpdc_ab_slg8_8.so libpdc_ab_slg4_13.so libperiodicity_hip.so; do echo "$lib"; PDC_LIBRARY=periodicity_amd/$lib SHAPES="50000x100000,40000x50000" python tools/sl_shapes.py; done
} > gpurun_out/r03o/sl.txt 2>&1
cat gpurun_out/r03o/sl.txt
