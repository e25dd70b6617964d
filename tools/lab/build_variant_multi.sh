#!/bin/bash
# build_variant_multi.sh <tag> "<src1.hip src2.hip ...>" [extra hipcc flags...]: like build_variant.sh with SEVERAL translation
# units recompiled under the extra flags (a switch that lives in a shared header, e.g. -DAVA_BUFLOAD=1 in conv_common.h)
set -e
tag=$1; srcs=$2; shift 2
cd "$(dirname "$0")/../../autoencoded-vocal-analysis_amd/csrc"
make -j8 > /dev/null
mkdir -p lab/obj
objs=""; skip=""
for src in $srcs; do
  obj=lab/obj/${src%.hip}_$tag.o
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wall -Wno-unused-function "$@" -c $src -o $obj &
  objs="$objs $obj"; skip="$skip|^${src%.hip}.o\$"
done
wait
others=$(ls *.o | grep -v -E "${skip#|}")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $objs -o libava_hip_$tag.so -Wl,-rpath,/opt/rocm/lib -lpthread
echo built libava_hip_$tag.so
