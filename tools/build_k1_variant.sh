#!/bin/bash
# tools/build_k1_variant.sh NAME "-DFLAG=.." [FILE] : build_variants/lib_NAME.so = the current objects with FILE (default k_integrate) recompiled with extra defines (A/B; RGBDR_LIB=... loads it)
set -e
cd "$(dirname "$0")/../rgbd-recon_amd/csrc"
NAME=$1; EXTRA=$2; FILE=${3:-k_integrate}
make -s
mkdir -p ../../build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-vectorize -fno-slp-vectorize -Wno-unused-result -Wno-unused-value $EXTRA -c $FILE.hip -o /tmp/${FILE}_$NAME.o
OBJS=$(ls *.o | grep -v "^$FILE.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../build_variants/lib_$NAME.so $OBJS /tmp/${FILE}_$NAME.o -ldl
echo built build_variants/lib_$NAME.so
