#!/bin/bash
# A/B aid: a variant of ONE translation unit linked with the other objects of the current build.
# usage: bash tools/build_variant.sh <name> <file.hip> <-D flags...>   -> vrpgym_hip/libvar_<name>.so
set -eu
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../vrp-gym_amd/csrc"
make -s -j8
base=$(basename $src .hip)
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-result -ffp-contract=on "$@" -c $src -o build/var_$name.o
objs=$(ls build/*.o | grep -v "build/var_" | grep -v "build/$base.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o ../vrpgym_hip/libvar_$name.so $objs build/var_$name.o
echo built libvar_$name.so
