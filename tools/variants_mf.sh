#!/bin/bash
# Developer helper (GPU box): time the list scan (both filters) with each build/librrtx_<name>.so named on the command line.
#   gpurun -- 'bash tools/variants_mf.sh base prio1'
cd $GRAFT_REPO_ROOT
cp rrt_amd/librrtx.so /tmp/librrtx_orig.so
for n in "$@"; do
  cp build/librrtx_$n.so rrt_amd/librrtx.so
  echo "== $n"
  timeout -k 10 120 python3 tools/mf_time.py 500 2>&1 | grep -v amdgpu.ids
done
cp /tmp/librrtx_orig.so rrt_amd/librrtx.so
