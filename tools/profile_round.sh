#!/bin/bash
# Developer helper (GPU box): the measurements behind profiles/<tag>_* — bench line (with its live PMC passes), kernel
# trace of the same command, and one rocprofv3 --pmc pass per counter group (never combined with tracing) for the four
# headline kernels: list scan / accelerated x fp32 / fp64, each on the `rrt` / `rrtd` binary rendering configuration 3.
#   gpurun -- 'bash tools/profile_round.sh r02'
set -e
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err || tail -3 $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pmc --no-configs > $O/bench_under_rocprof.json 2> $O/trace.err || tail -3 $O/trace.err
ARGS="-i $R/scenes/final.txt -w 1200 -h 800 -s 500 -d 50"
for variant in list_f32 accel_f32 list_f64 accel_f64; do
  case $variant in
    list_f32)  EXE=$R/rrt;  EXTRA="-b" ;;
    accel_f32) EXE=$R/rrt;  EXTRA="" ;;
    list_f64)  EXE=$R/rrtd; EXTRA="-b" ;;
    accel_f64) EXE=$R/rrtd; EXTRA="" ;;
  esac
  i=0
  for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SMEM SQ_LDS_IDX_ACTIVE" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_COEXEC_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout -k 10 120 rocprofv3 --output-format csv --pmc $grp -d $O/$variant/p$i -o p$i -- $EXE $ARGS $EXTRA -o $O/$variant.png > /dev/null 2> $O/$variant.p$i.err || tail -3 $O/$variant.p$i.err
  done
done
cd $R
python3 - "$O" "$TAG" <<'PY'
import json, sys
o, tag = sys.argv[1], sys.argv[2]
b = json.load(open(o + "/bench.json"))
r = b["roofline"]
live = {"kernel_source_sha": r.get("kernel_source_sha"), "list_scan": dict(r.get("counters", {})), "note": "written by tools/profile_round.sh from the bench line of the same run (profiles/%s_bench.json): its live PMC passes; "
        "bench.py falls back to this file only when it cannot profile and the device sources hash to kernel_source_sha" % tag}
if r.get("mfma", {}).get("counters"):
    live["list_scan_mfma"] = dict(r["mfma"]["counters"])
if r.get("traffic_detail"):
    live["hbm_read_bytes"], live["hbm_write_bytes"] = r["traffic_detail"]["read_bytes"], r["traffic_detail"]["write_bytes"]
json.dump(live, open(o + "/pmc_live.json", "w"), indent=1)
PY
python3 tools/pmc_report.py $O > $O/pmc_report.csv
cat $O/trace/*kernel_stats.csv | head -14
cat $O/pmc_report.csv
