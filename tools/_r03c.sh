mkdir -p gpurun_out/r03c
for s in 0 2 3 4; do echo "PDC_SL_SLICES=$s"; PDC_SL_SLICES=$s SHAPES="50000x100000,25000x100000" python tools/sl_shapes.py; done > gpurun_out/r03c/sl_slices.txt 2>&1
python -m pytest tests/test_multi_gpu.py -x -q 2>&1 | tail -3 >> gpurun_out/r03c/sl_slices.txt
cat gpurun_out/r03c/sl_slices.txt
