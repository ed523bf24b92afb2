mkdir -p gpurun_out/r03j
{
for lib in libperiodicity_hip.so libpdc_ab_chunk256.so libpdc_ab_chunk64.so libperiodicity_hip.so libpdc_ab_chunk256.so; do echo "$lib"; PDC_LIBRARY=periodicity_amd/$lib SHAPES="50000x100000,50000x280000,200000x20000" python tools/pdm_shapes.py; done
} > gpurun_out/r03j/pdm.txt 2>&1
cat gpurun_out/r03j/pdm.txt
