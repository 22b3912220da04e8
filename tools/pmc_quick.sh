#!/bin/bash
# Developer helper (GPU box): three PMC passes of `rrt <args>` and the derived utilisations of its kernels.
#   gpurun -- 'bash tools/pmc_quick.sh accel_f32 -s 500'        (variant name = a directory tools/pmc_report.py knows)
set -e
VAR=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcq_$VAR
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
EXE=$R/rrt
case $VAR in *f64) EXE=$R/rrtd ;; esac
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --output-format csv --pmc $grp -d $O/$VAR/p$i -o p$i -- $EXE -i $R/scenes/final.txt -w 1200 -h 800 -d 50 "$@" -o $O/frame.png > /dev/null 2> $O/p$i.err || tail -3 $O/p$i.err
done
cd $R
python3 tools/pmc_report.py $O | grep -v "^#   \|^# values"
