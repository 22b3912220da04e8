#!/bin/bash
# Developer helper (GPU box): PMC passes for one kernel of a short render.  usage: tools/pmc_tail.sh W H spp
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_tail
rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --output-format csv --pmc $grp -d $O/p$i -o p$i -- python3 $R/tools/one_render.py $1 $2 $3 2 > $O/log$i.txt 2>&1 || { tail -5 $O/log$i.txt; }
done
cd $R
for k in tail_kernel render_kernel; do echo "== $k"; PMC_KERNEL=$k python3 tools/pmc_summary.py $O/p1 $O/p2 $O/p3 $O/p4; done
