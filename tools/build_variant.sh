#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG=.. -DFLAG2=.." : builds build_variants/lib_NAME.so with extra defines (A/B experiments; use RGBDR_LIB=... to load it)
set -e
cd "$(dirname "$0")/../rgbd-recon_amd/csrc"
NAME=$1; EXTRA=$2
T=$(mktemp -d)
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-vectorize -fno-slp-vectorize -Wno-unused-result -Wno-unused-value $EXTRA"
mkdir -p ../../build_variants
for f in $(ls *.hip | sed 's/\.hip$//'); do /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS -c $f.hip -o $T/$f.o & done
for f in abi comm file_io calib_inverter; do /opt/rocm/bin/hipcc $FLAGS -c $f.cpp -o $T/$f.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../build_variants/lib_$NAME.so $T/*.o -ldl
rm -rf $T
echo built build_variants/lib_$NAME.so
