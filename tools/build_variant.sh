#!/bin/bash
# A library variant for A/B timing on the box: ONE translation unit recompiled with extra flags, linked with the in-tree objects of the rest.
#   tools/build_variant.sh NAME UNIT [flags...]     e.g.  tools/build_variant.sh nolog extrack_rev -DXT_REV_DIAG=1
# -> extrack_amd/var_NAME.so (git-ignored; travels to the box); use with EXTRACK_HIP_LIB=$PWD/extrack_amd/var_NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; unit=$2; shift 2
flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xclang -target-feature -Xclang -load-store-opt"
hipcc $flags "$@" -c extrack_amd/csrc/$unit.hip -o build/var_${name}_$unit.o
objs=""
for o in build/extrack_*.o; do
  [ "$o" = "build/$unit.o" ] && objs="$objs build/var_${name}_$unit.o" || objs="$objs $o"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o extrack_amd/var_$name.so $objs
echo extrack_amd/var_$name.so
