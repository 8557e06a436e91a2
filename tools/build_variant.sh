#!/bin/bash
# usage: tools/build_variant.sh NAME SOURCE.hip -DFLAG...   -> mafed_amd/lib_NAME.so (one source rebuilt with extra flags)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
python mafed_amd/build.py > /dev/null
mkdir -p /tmp/mafed_var
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DNDEBUG "$@" -c mafed_amd/csrc/$src.hip -o /tmp/mafed_var/$name.o
objs=$(ls mafed_amd/_build/*.o | grep -v "/$src.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o mafed_amd/lib_$name.so $objs /tmp/mafed_var/$name.o
echo mafed_amd/lib_$name.so
