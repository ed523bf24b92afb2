mkdir -p gpurun_out/r03f
python -m pytest tests/test_phase_gpu.py tests/test_random_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r03f/tests.txt
{
echo "r02 library"; PDC_LIBRARY=periodicity_amd/libpdc_ab_r02.so SHAPES="50000x100000,25000x100000,10000x20000,2000x100000,500x100000,2000x1000" python tools/sl_shapes.py
echo "now"; SHAPES="50000x100000,25000x100000,10000x20000,2000x100000,500x100000,2000x1000" python tools/sl_shapes.py
} > gpurun_out/r03f/ab.txt 2>&1
cat gpurun_out/r03f/tests.txt gpurun_out/r03f/ab.txt
