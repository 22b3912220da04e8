#!/bin/bash
# Developer helper (GPU box): list scan and accelerated mode at spp 500 with each build/librrtx_<name>.so named on the command line.
#   gpurun -- 'bash tools/variants_both.sh maxilp bias0'
cd $GRAFT_REPO_ROOT
cp rrt_amd/librrtx.so /tmp/librrtx_orig.so
for n in "$@"; do
  cp build/librrtx_$n.so rrt_amd/librrtx.so
  echo "== $n"
  timeout -k 10 150 python3 tools/accel_time.py 500 2>&1 | grep -v amdgpu.ids
done
cp /tmp/librrtx_orig.so rrt_amd/librrtx.so
