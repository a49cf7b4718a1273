#!/usr/bin/env python3
"""GPU timeline around one kernel from a rocprofv3 kernel-trace database: for the non-empty dispatches of kernels whose name contains
NEEDLE, the median / max of (a) the kernel's own duration, (b) the idle gap between the end of the previous dispatch and its start,
(c) the gap between its end and the start of the next dispatch, and the names of those neighbours.
usage: rocpd_timeline.py results.db [needle]"""
import sqlite3, statistics, sys

db = sqlite3.connect(sys.argv[1])
needle = sys.argv[2] if len(sys.argv) > 2 else 'env_trace_kernel<false>'
cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
rows = db.execute('select name, start, end from kernels order by start').fetchall()
idx = [i for i, r in enumerate(rows) if needle in r[0]]
if not idx:
    sys.exit('no dispatch of %s' % needle)
dmax = max(rows[i][2] - rows[i][1] for i in idx)
real = [i for i in idx if rows[i][2] - rows[i][1] >= 0.1 * dmax]
dur = [(rows[i][2] - rows[i][1]) / 1e6 for i in real]
before = [(rows[i][1] - rows[i - 1][2]) / 1e6 for i in real if i > 0]
after = [(rows[i + 1][1] - rows[i][2]) / 1e6 for i in real if i + 1 < len(rows)]
prev = {}
for i in real:
    if i > 0:
        prev[rows[i - 1][0][:40]] = prev.get(rows[i - 1][0][:40], 0) + 1
f = lambda v: '%.3f / %.3f' % (statistics.median(v), max(v)) if v else 'n/a'
print('%s: %d non-empty dispatches of %d | duration ms median/max %s | idle gap before %s | gap after %s | previous kernel: %s'
      % (needle, len(real), len(idx), f(dur), f(before), f(after), prev))
