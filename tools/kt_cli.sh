#!/bin/bash
# Developer helper (GPU box): rocprofv3 --kernel-trace --stats of one `rrt` run; args after the tag go to rrt.   gpurun -- 'bash tools/kt_cli.sh tag -s 500 -C -1 -b'
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_$TAG -o kt -- $R/rrt -i $R/scenes/final.txt -w 1200 -h 800 "$@" -o /tmp/x.png > /dev/null 2>&1
f=$(find $R/gpurun_out/kt_$TAG -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && python3 -c "import csv,sys; [print(\"%-72s %3s %10.3f ms\" % (r[\"Name\"][:72], r[\"Calls\"], float(r[\"TotalDurationNs\"]) / 1e6)) for r in list(csv.DictReader(open(sys.argv[1])))[:5]]" "$f"
