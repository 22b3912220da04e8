#!/bin/bash
# Developer script (GPU box, -DRRTX_EXPERIMENTS build over rrt_amd/librrtx.so): the mesh scene over the grid's cell size
# (multiples of the median primitive extent) and the limit on the number of cells.
for cell in 2 3 4 6 8 12; do
  for maxc in 2097152 262144; do
    echo "RRTX_GRID_CELL=$cell RRTX_GRID_MAXCELLS=$maxc"
    RRTX_GRID_CELL=$cell RRTX_GRID_MAXCELLS=$maxc timeout -k 5 60 python tools/mesh_trace.py 0 | tail -1
  done
done
