#!/bin/bash
# Developer helper (GPU box): kernel traces of the launches that are all tail - C2 (test1.txt spp 10), shard 3 of 8 of C3 - in both modes; every kernel with its
# calls and total time.  (rrt renders once per process: calls = kernels per launch.)   gpurun -- 'bash tools/kt_ends.sh tag'
TAG=${1:-x}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { # name, rrt args...
  n=$1; shift
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_${TAG}_$n -o kt -- $R/rrt -w 1200 -h 800 "$@" -o /tmp/x.png > /dev/null 2>&1
  f=$(find $R/gpurun_out/kt_${TAG}_$n -name '*kernel_stats.csv' | head -1)
  echo "== $n"
  [ -n "$f" ] && python3 -c "import csv,sys; [print(\"%-80s %3s %10.3f ms\" % (r[\"Name\"][:80], r[\"Calls\"], float(r[\"TotalDurationNs\"]) / 1e6)) for r in list(csv.DictReader(open(sys.argv[1])))[:9]]" "$f"
}
run c2  -i $R/scenes/test1.txt -s 10 -b
run c2a -i $R/scenes/test1.txt -s 10
run sh  -i $R/scenes/final.txt -s 500 -b -R 3 -N 8 -T 4
run sha -i $R/scenes/final.txt -s 500 -R 3 -N 8 -T 4
