#!/bin/bash
# Run on the GPU box (through gpurun): the streamed StringLength kernels against the oracle, then their timing
# against the other kernels (tools/sl_stream_gpu.sh [check|time|prof]).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mode=${1:-all}
if [ "$mode" = all ] || [ "$mode" = check ]; then
  PDC_SL_STREAM_MIN=4096 timeout 600 python3 tools/sl_stream_check.py 70000x48 4097x33 20000x40e 150001x12 9000x20e 400000x6 2>&1 | tail -8
fi
if [ "$mode" = all ] || [ "$mode" = time ]; then
  export SHAPES=${SHAPES:-1000000x2048,400000x2048,250000x4096,200000x8192,74326x20000}
  echo "== default dispatch"; PDC_SL_STREAM_DEBUG=1 python3 tools/sl_shapes.py 2>&1 | grep -E "N=|left" | sort | uniq -c
  echo "== no streamed kernels"; PDC_SL_STREAM=0 python3 tools/sl_shapes.py 2>&1 | grep "N="
fi
