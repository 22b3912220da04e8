#!/bin/bash
# Developer helper (GPU box): HBM bytes per launch of the list-scan kernel (FETCH_SIZE / WRITE_SIZE passes) + kernel times.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/traffic; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 120 rocprofv3 --output-format csv --pmc $c -d $O/$c -o p -- $R/rrt -b -i $R/scenes/final.txt -w 1200 -h 800 -s 500 -o $O/x.png > /dev/null 2>&1
done
python3 - <<PY
import csv, glob
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = {}
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if "rrtx" in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("rrtx::")[1][:48]
                acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"])
    for k, v in acc.items():
        print("%-12s %-50s %.3f GB" % (c, k, v * 1024 * (2 if c == "FETCH_SIZE" else 1) / 1e9))
PY
cd $R && python3 tools/accel_time.py 48 500 2>&1 | grep f32
