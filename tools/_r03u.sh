mkdir -p gpurun_out/r03u
{
python -m pytest tests/test_aov_ce_gpu.py tests/test_phase_gpu.py tests/test_multi_gpu.py -x -q 2>&1 | tail -4
echo "PDC_PDM_SPLIT=0 (unsplit)"; PDC_PDM_SPLIT=0 python tools/ce_gl_timing.py
echo "default"; python tools/ce_gl_timing.py
} > gpurun_out/r03u/ce.txt 2>&1
cat gpurun_out/r03u/ce.txt
