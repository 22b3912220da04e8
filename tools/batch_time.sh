#!/bin/bash
# Developer helper (GPU box): wall time per frame of a CLI batch at the reference's animation settings
# (scenes/final_anim/Makefile: rrt -s 50 -w 1280 -h 720), N copies of final.txt in one process.
#   batch_time.sh [N] [spp] [extra rrt flags]
R=$GRAFT_REPO_ROOT; N=${1:-60}; SPP=${2:-50}; shift; shift; O=/tmp/batch_out; rm -rf $O; mkdir -p $O
args=""
for i in $(seq 1 $N); do args="$args -i $R/scenes/final.txt -o $O/f$i.png"; done
s=$(date +%s.%N)
$R/rrt -s $SPP -w 1280 -h 720 "$@" $args 2> $O/err.txt > /dev/null
e=$(date +%s.%N)
python3 -c "
import re
t=[float(x) for x in re.findall(r'took ([0-9.e+-]+) seconds', open('$O/err.txt').read())]
print('%d frames spp $SPP [$*]: %.1f ms wall, %.3f ms per frame, render %.3f ms per frame' % ($N, ($e-$s)*1e3, ($e-$s)*1e3/$N, sum(t)*1e3/len(t)))"
