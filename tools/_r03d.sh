mkdir -p gpurun_out/r03d
python -m pytest tests/test_phase_gpu.py -x -q -k "stringlength or phase_scans or period_grid" 2>&1 | tail -15 > gpurun_out/r03d/tests.txt
for d in 0 1; do echo "PDC_SL_DUO=$d"; PDC_SL_DUO=$d SHAPES="50000x100000,25000x100000,10000x20000,2000x100000,500x100000" python tools/sl_shapes.py; done > gpurun_out/r03d/sl_duo.txt 2>&1
cat gpurun_out/r03d/tests.txt gpurun_out/r03d/sl_duo.txt
