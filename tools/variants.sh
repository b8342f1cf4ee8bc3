#!/bin/bash
# tools/variants.sh NAME "-DFLAG1 -DFLAG2"   → builds optixpathtracer_amd/variants/libptamd_NAME.so (A/B experiments)
set -e
NAME=$1; DEFS=$2
cd "$(dirname "$0")/../optixpathtracer_amd/csrc"
mkdir -p ../variants
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -Wno-unused-result -Wno-unused-value $DEFS"
/opt/rocm/bin/hipcc $FLAGS -c pt_api.hip -o /tmp/pt_api_$NAME.o &
/opt/rocm/bin/hipcc $FLAGS -c pt_bvh_build.hip -o /tmp/pt_bvh_build_$NAME.o &
wait
make -s pt_objload.o  # the host-only scene ingestion is the same in every variant
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libptamd_$NAME.so /tmp/pt_api_$NAME.o /tmp/pt_bvh_build_$NAME.o pt_objload.o -ldl
echo built variants/libptamd_$NAME.so
