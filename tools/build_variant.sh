#!/bin/bash
# A/B build of the library with extra -D flags on ONE source file (the other objects are reused):
#   tools/build_variant.sh <name> <file.hip> -DFOO=1 ...   ->  inferbiomechanics_amd/lib/ab/libib_hip_<name>.so
# Use with IB_HIP_LIB=<that path> (hip.py) to run both builds inside one gpurun call (same box, same clocks).
set -e
NAME=$1; SRC=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/inferbiomechanics_amd/csrc
make -s -C $C -j8 >/dev/null
mkdir -p $C/build_ab/var $ROOT/inferbiomechanics_amd/lib/ab
# variants are measurement builds: -DIB_AB objects (A/B switches, stamp hooks) + the one re-compiled source
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -DIB_AB "$@" -c $C/$SRC -o $C/build_ab/var/${SRC%.hip}_$NAME.o
OBJS=$(ls $C/build_ab/*.o | grep -v "/${SRC%.hip}.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/inferbiomechanics_amd/lib/ab/libib_hip_$NAME.so $OBJS $C/build_ab/var/${SRC%.hip}_$NAME.o
echo $ROOT/inferbiomechanics_amd/lib/ab/libib_hip_$NAME.so
