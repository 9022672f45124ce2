"""python examples/trace_timeline.py <rocprofv3 results .db> [first_collide_index] [n_kernels]

Prints the kernel timeline of a rocprofv3 --kernel-trace run (rocpd sqlite output): start offset, duration, gap to
the previous kernel end on any stream, stream id, name.  Used to find idle time between the phases of a step."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 80
rows = db.execute("select start, end, stream_id, name, grid_x, grid_y from kernels order by start").fetchall()
t0 = rows[skip][0]
busy_end = rows[skip][0]
for start, end, stream, name, gx, gy in rows[skip:skip + count]:
    gap = (start - busy_end) / 1e3
    busy_end = max(busy_end, end)
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0][:40]
    print("%10.1f us  dur %8.1f  gap %7.1f  s%-2d %-40s grid %d x %d" % ((start - t0) / 1e3, (end - start) / 1e3, gap, stream, short, gx, gy))
