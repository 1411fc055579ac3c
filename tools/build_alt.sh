#!/bin/bash
# Builds tools/_alt/lib_<tag>.so: the library with ONE source recompiled under extra flags (timing ablations / A-B runs; select it
# with MS_LIB_PATH).   usage: tools/build_alt.sh TAG source.hip -DFLAG=1 ...
set -e
cd "$(dirname "$0")/../mix_stage_amd/csrc"
TAG=$1; SRC=$2; shift 2
mkdir -p ../../tools/_alt
OBJ=../../tools/_alt/${SRC%.hip}_$TAG.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wall -Wno-unused-function "$@" -c $SRC -o $OBJ
OBJS=$(ls *.o | grep -v "^${SRC%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_alt/lib_$TAG.so $OBJS $OBJ
rm -f $OBJ
ls -la ../../tools/_alt/lib_$TAG.so
