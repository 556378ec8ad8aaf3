#!/bin/bash
# tools/build_variant_perfile.sh NAME "DEFAULT_EXTRA" "file1:EXTRA1" "file2:EXTRA2" ... : like build_variant.sh with per-file extra flags
set -e
cd "$(dirname "$0")/../rgbd-recon_amd/csrc"
NAME=$1; DEF=$2; shift 2
T=$(mktemp -d)
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-vectorize -fno-slp-vectorize -Wno-unused-result -Wno-unused-value"
mkdir -p ../../build_variants
for f in $(ls *.hip | sed 's/\.hip$//'); do
  X=$DEF
  for spec in "$@"; do if [ "${spec%%:*}" = "$f" ]; then X=${spec#*:}; fi; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE $X -c $f.hip -o $T/$f.o &
done
for f in abi file_io calib_inverter; do /opt/rocm/bin/hipcc $BASE -c $f.cpp -o $T/$f.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../build_variants/lib_$NAME.so $T/*.o
rm -rf $T
echo built build_variants/lib_$NAME.so
