#!/bin/bash
# Developer helper (GPU box): the measurements behind profiles/r01_* — bench line, kernel trace, PMC passes.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_final
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err || tail -3 $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/trace.err || tail -3 $O/trace.err
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SMEM SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --output-format csv --pmc $grp -d $O/p$i -o p$i -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc$i.json 2> $O/pmc$i.err || tail -3 $O/pmc$i.err
done
cd $R
for k in "render_kernel<float, true, 1, false, 0, false>" "render_kernel<float, true, 0, false, 2, false>" "render_kernel<float, true, 0, false, 2, true>" tail_kernel; do echo "== $k"; PMC_KERNEL="$k" python3 tools/pmc_summary.py $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 $O/p6 $O/p7 $O/p8; done > $O/pmc_summary.txt
cat $O/trace/*kernel_stats.csv | head -12
cat $O/pmc_summary.txt
