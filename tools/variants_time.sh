#!/bin/bash
# Developer helper (GPU box): time the accelerated mode with each build/librrtx_<name>.so named on the command line.
#   gpurun -- 'bash tools/variants_time.sh base s2 s1'
cd $GRAFT_REPO_ROOT
cp rrt_amd/librrtx.so /tmp/librrtx_orig.so
for n in "$@"; do
  cp build/librrtx_$n.so rrt_amd/librrtx.so
  echo "== $n"
  timeout -k 10 120 python3 tools/accel_time.py 48 500 2>&1 | grep -v amdgpu.ids | grep f32
done
cp /tmp/librrtx_orig.so rrt_amd/librrtx.so
