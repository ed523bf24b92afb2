#!/bin/bash
# Build an experimental variant of the HIP library next to the product one:
#   tools/ab_build.sh NAME "-DPDC_AB=1" [file.hip]   ->  periodicity_amd/libpdc_ab_NAME.so
# then compare on one GPU box:  PDC_LIBRARY=periodicity_amd/libpdc_ab_NAME.so python tools/bench_configs.py ...
set -e
cd "$(dirname "$0")/../periodicity_amd/csrc"
name=$1; flags=$2; src=${3:-gls.hip}
make -s
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $flags -c $src -o /tmp/ab_$name.o
objs=$(ls *.o | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 $objs /tmp/ab_$name.o -shared -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -o ../libpdc_ab_$name.so
echo built periodicity_amd/libpdc_ab_$name.so
