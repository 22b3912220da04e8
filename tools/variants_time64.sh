#!/bin/bash
# Developer helper (GPU box): like variants_time.sh, fp64 lines.
cd $GRAFT_REPO_ROOT
cp rrt_amd/librrtx.so /tmp/librrtx_orig.so
for n in "$@"; do
  [ "$n" != "cur" ] && cp build/librrtx_$n.so rrt_amd/librrtx.so
  echo "== $n"
  timeout -k 10 120 python3 tools/accel_time.py 48 500 2>&1 | grep f64
  cp /tmp/librrtx_orig.so rrt_amd/librrtx.so
done
