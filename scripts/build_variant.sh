#!/bin/bash
# development: build eas_snn_amd/libeas_exp_<tag>.so from the current objects with ONE source recompiled with extra flags
# usage: build_variant.sh <tag> <source.hip> <extra flags...>     (select at run time with EAS_LIB=<path>)
set -e
TAG=$1; SRC=$2; shift 2
cd "$(dirname "$0")/../eas_snn_amd/csrc"
make -s -j8 > /dev/null
OBJ=/tmp/eas_variant_${TAG}_$(basename $SRC .hip).o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -Wall -Wno-unused-function "$@" -c $SRC -o $OBJ
OTHERS=$(ls *.o | grep -v "^$(basename $SRC .hip).o$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt $OTHERS $OBJ -o ../libeas_exp_${TAG}.so
echo built ../libeas_exp_${TAG}.so
