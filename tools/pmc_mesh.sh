#!/bin/bash
# Developer helper (GPU box): PMC passes of `rrt` / `rrtd` on the 27 072-triangle mesh scene of tests/test_gpu_mesh.py
# (600 x 400, spp 16) - the scene class whose grid lives in HBM -> profiles/r02_pmc_mesh.csv
#   gpurun -- 'bash tools/pmc_mesh.sh'
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_mesh
rm -rf $O; mkdir -p $O
python3 -c "
import sys; sys.path.insert(0, '$R/tests')
from _oracle import mesh_scene
print(mesh_scene('$O/mesh27k.txt', 48, 96))"
cd /tmp && export TMPDIR=/tmp
for variant in accel_f32 accel_f64; do
  case $variant in
    accel_f32) EXE=$R/rrt ;;
    accel_f64) EXE=$R/rrtd ;;
  esac
  i=0
  for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout -k 10 120 rocprofv3 --output-format csv --pmc $grp -d $O/$variant/p$i -o p$i -- $EXE -i $O/mesh27k.txt -w 600 -h 400 -s 16 -d 50 -o $O/$variant.png > /dev/null 2> $O/$variant.p$i.err || tail -3 $O/$variant.p$i.err
  done
done
cd $R
python3 tools/pmc_report.py $O | sed 's/final.txt -w 1200 -h 800 -s 500/<27 072-triangle mesh> -w 600 -h 400 -s 16/' > $O/pmc_mesh.csv
grep "^# rrt\|valu_issue_utilisation\|valu_lane_utilisation\|hbm_bandwidth\|hbm_read" $O/pmc_mesh.csv
