#!/bin/bash
# Developer script (GPU box, -DRRTX_EXPERIMENTS build over rrt_amd/librrtx.so): the mesh scene over the grid's cell size
# (multiples of the median primitive extent).
for cell in 1.0 1.25 1.5 1.75 2 2.5; do
    echo "RRTX_GRID_CELL=$cell"
    RRTX_GRID_CELL=$cell timeout -k 5 60 python tools/mesh_trace.py 0 | tail -1
    RRTX_GRID_CELL=$cell timeout -k 5 60 python tools/mesh_trace.py 1 | tail -1
    RRTX_GRID_CELL=$cell timeout -k 5 60 python tools/mesh_trace.py 0 16 32 | tail -1
done
