mkdir -p gpurun_out/r03v
{
python -m pytest tests/test_aov_ce_gpu.py tests/test_phase_gpu.py tests/test_multi_gpu.py tests/test_random_gpu.py -x -q 2>&1 | tail -6
python tools/ce_gl_timing.py
SHAPES="50000x100000" python tools/pdm_shapes.py
} > gpurun_out/r03v/ce.txt 2>&1
cat gpurun_out/r03v/ce.txt
