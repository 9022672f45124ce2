#!/bin/bash
out=$PWD/gpurun_out/r3l; mkdir -p $out; root=$PWD
export TMPDIR=/tmp
cd $root
rocprofv3 --kernel-trace -d $out/trace -o t -- python3 examples/rccl_selfloop.py 256 60 rccl ${NXT:-64} > $out/trace.txt 2>&1 || { tail -5 $out/trace.txt; exit 1; }
db=$(find $out/trace -name "*.db" | head -1)
python3 - <<PY
import sqlite3
db = sqlite3.connect("$db")
rows = db.execute("select start, end, stream_id, name, grid_x, grid_y from kernels order by start").fetchall()
print(len(rows), "kernels")
# the slab phase: last ~ 60*? kernels; print a window near the end covering ~7 steps
import sys
n=len(rows)
w=rows[n-260:n-60]
t0=w[0][0]; busy=w[0][0]
for s,e,st,name,gx,gy in w:
    gap=(s-busy)/1e3; busy=max(busy,e)
    short=name.replace("(anonymous namespace)::","").replace("void ","").split("(")[0].split("<")[0][:38]
    print("%9.1f us dur %7.1f gap %6.1f s%-2d %-38s %dx%d"%((s-t0)/1e3,(e-s)/1e3,gap,st,short,gx,gy))
PY
rm -rf $out/trace
