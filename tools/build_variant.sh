#!/bin/bash
# Developer helper: build a variant of the library into the git-ignored build/ directory.
#   tools/build_variant.sh <name> [extra hipcc -D flags...]     ->  build/librrtx_<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build
K="--offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form -fPIC -std=c++17 $RRTX_KFLAGS"
/opt/rocm/bin/hipcc $K "$@" -c rrt_amd/csrc/rrtx_kernels.hip -o build/k_$name.o
/opt/rocm/bin/hipcc -O2 -ffp-contract=off -fPIC -std=c++17 "$@" -c rrt_amd/csrc/rrtx_api.cpp -o build/api_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/librrtx_$name.so build/k_$name.o build/api_$name.o rrt_amd/csrc/rrtx_group.o rrt_amd/csrc/host_scene.o rrt_amd/csrc/host_image.o -lz -ldl
echo build/librrtx_$name.so
