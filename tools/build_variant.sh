#!/bin/bash
# Builds a variant of the HIP library beside the shipped one, for same-box A/B runs (tools/ab.sh):
#   tools/build_variant.sh <name> [extra hipcc flags, e.g. -DPBD_DT_APPROX=0]   -> partsbaseddetector_amd/csrc/build/libpbd_<name>.so
# Only the sources are recompiled with the extra flags; objects go to a directory of their own.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
obj=$R/partsbaseddetector_amd/csrc/build/var_$name
mkdir -p $obj
flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -w"
for s in pbd_capi pbd_kernels_features pbd_kernels_conv pbd_kernels_conv_mfma pbd_kernels_dp; do
  hipcc $flags "$@" -c -o $obj/$s.o $R/partsbaseddetector_amd/csrc/$s.hip &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/partsbaseddetector_amd/csrc/build/libpbd_$name.so $obj/*.o
echo $R/partsbaseddetector_amd/csrc/build/libpbd_$name.so
