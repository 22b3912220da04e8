#!/bin/bash
# Developer helper (GPU box): kernel trace (rocprofv3 --kernel-trace --stats) of `rrt` on final.txt 1200x800 spp $1 in both modes -> gpurun_out/kt_<tag>_{b,a}
#   gpurun -- 'bash tools/kt_modes.sh 500 tag'
SPP=${1:-500}; TAG=${2:-x}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for m in b a; do
  if [ $m = b ]; then F=-b; else F=; fi
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_${TAG}_$m -o kt -- $R/rrt -i $R/scenes/final.txt -w 1200 -h 800 -s $SPP $F -o /tmp/x.png > /dev/null 2>&1
  f=$(find $R/gpurun_out/kt_${TAG}_$m -name '*kernel_stats.csv' | head -1)
  echo "== mode $m"
  [ -n "$f" ] && python3 -c "import csv,sys; [print(\"%-72s %3s %10.3f ms\" % (r[\"Name\"][:72], r[\"Calls\"], float(r[\"TotalDurationNs\"]) / 1e6)) for r in list(csv.DictReader(open(sys.argv[1])))[:7]]" "$f"
done
