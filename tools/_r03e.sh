mkdir -p gpurun_out/r03e
{
echo "r02 library"; PDC_LIBRARY=periodicity_amd/libpdc_ab_r02.so SHAPES="50000x100000,25000x100000" python tools/sl_shapes.py
echo "now, PDC_SL_DUO=0"; PDC_SL_DUO=0 SHAPES="50000x100000,25000x100000" python tools/sl_shapes.py
echo "now, duo"; SHAPES="50000x100000,25000x100000" python tools/sl_shapes.py
echo "r02 library again"; PDC_LIBRARY=periodicity_amd/libpdc_ab_r02.so SHAPES="50000x100000,25000x100000" python tools/sl_shapes.py
} > gpurun_out/r03e/ab.txt 2>&1
cat gpurun_out/r03e/ab.txt
