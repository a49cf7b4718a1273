#!/usr/bin/env python3
"""One steady-state iteration from a rocprofv3 kernel-trace database as a timeline: every dispatch between two consecutive launches of
the ANCHOR kernel (default: the light-probe pdf update, the first launch of an iteration) with its start offset, duration, queue and the
idle time of the device before it (no dispatch running on any queue).  usage: rocpd_iteration.py results.db [anchor] [which]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
anchor = sys.argv[2] if len(sys.argv) > 2 else 'light_rows_kernel'
which = int(sys.argv[3]) if len(sys.argv) > 3 else -3
cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
qcol = 'queue_id' if 'queue_id' in cols else ('queue' if 'queue' in cols else ('stream_id' if 'stream_id' in cols else None))
rows = db.execute('select name, start, end%s from kernels order by start' % (', ' + qcol if qcol else '')).fetchall()
marks = [i for i, r in enumerate(rows) if anchor in r[0]]
# an iteration may launch the anchor kernel more than once in a row: keep the first of each burst (> 1 ms apart)
first = [m for k, m in enumerate(marks) if k == 0 or rows[m][1] - rows[marks[k - 1]][1] > 1e6]
if len(first) < 4:
    sys.exit('fewer than four iterations of %s in the trace (columns: %s)' % (anchor, cols))
a, b = first[which], first[which + 1]
t0 = rows[a][1]
print('iteration of %.3f ms, %d dispatches, queue column: %s' % ((rows[b][1] - t0) / 1e6, b - a, qcol))
print('%9s %9s %9s  %-6s %s' % ('start us', 'dur us', 'idle us', 'queue', 'kernel'))
busy_until = t0
for r in rows[a:b]:
    idle = max(0, r[1] - busy_until)
    print('%9.1f %9.1f %9.1f  %-6s %s' % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, idle / 1e3, r[3] if qcol else '-', r[0][:90]))
    busy_until = max(busy_until, r[2])
