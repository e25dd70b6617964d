#!/bin/bash
# build_variant.sh <tag> <source.hip> [extra hipcc flags...]: libava_hip_<tag>.so = the product objects with ONE translation
# unit recompiled under extra flags (e.g. -DAVA_FL_CFG="8,4,4,4,3"); load it with AVA_HIP_LIB_TAG=<tag> for a same-box A/B.
set -e
tag=$1; src=$2; shift 2
cd "$(dirname "$0")/../../autoencoded-vocal-analysis_amd/csrc"
make -j8 > /dev/null
mkdir -p lab/obj
obj=lab/obj/${src%.hip}_$tag.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wall -Wno-unused-function "$@" -c $src -o $obj
others=$(ls *.o | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $obj -o libava_hip_$tag.so -Wl,-rpath,/opt/rocm/lib -lpthread
echo built libava_hip_$tag.so
