mkdir -p gpurun_out/r03g
{
echo "r02 library"; PDC_LIBRARY=periodicity_amd/libpdc_ab_r02.so SHAPES="50000x100000,50000x280000,200000x20000,50000x1000" python tools/pdm_shapes.py
for z in 0 1 2 3 5 8; do echo "PDC_PDM_NZ=$z"; PDC_PDM_NZ=$z SHAPES="50000x100000,50000x280000,200000x20000,50000x1000" python tools/pdm_shapes.py; done
} > gpurun_out/r03g/pdm.txt 2>&1
python -m pytest tests/test_phase_gpu.py tests/test_aov_ce_gpu.py -x -q -k "pdm or aov or phase_scans" 2>&1 | tail -5 >> gpurun_out/r03g/pdm.txt
cat gpurun_out/r03g/pdm.txt
