mkdir -p gpurun_out/r03i
{
for nb in 5 4 3; do for z in 1 3 4 5; do echo "NB=$nb NZ=$z"; NB=$nb PDC_PDM_NZ=$z SHAPES="50000x100000" python tools/pdm_shapes.py; done; done
} > gpurun_out/r03i/pdm.txt 2>&1
cat gpurun_out/r03i/pdm.txt
