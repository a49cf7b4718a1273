#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel trace) as the --stats table: per kernel calls, total,
average, min, max (us) and share; plus register/LDS use.  usage: rocpd_summary.py results.db [out.md]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("""select name, count(*), sum(duration), avg(duration), min(duration), max(duration),
                      max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size),
                      max(grid_x), max(workgroup_x)
                     from kernels group by name order by sum(duration) desc""").fetchall()
tot = sum(r[2] for r in rows) or 1
# the median is the steady-state figure: the average of a kernel includes its first launches, which pay the first touch
# of freshly allocated scratch (a handful of 5-ms outliers for the traversal kernel)
# 'real': dispatches lasting >= 10 % of the kernel's longest -- an env-shade launch issues its kernels once per chunk of the ray
# stream and the chunks behind the covered-pixel count are empty ~4-us dispatches that would drown the average
lines = ['| kernel | calls | total us | avg us | real calls | real avg us | real median us | min us | max us | % | vgpr | agpr | sgpr | lds B | scratch B | grid_x | wg_x |',
         '|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|']
for r in rows:
    d = [x[0] for x in db.execute('select duration from kernels where name = ? order by duration', (r[0],)).fetchall()]
    real = [x for x in d if x >= 0.1 * d[-1]] if d else []
    med = real[len(real) // 2] if real else 0
    ravg = sum(real) / len(real) if real else 0
    name = r[0] if len(r[0]) < 110 else r[0][:107] + '...'
    lines.append('| %s | %d | %.1f | %.2f | %d | %.2f | %.2f | %.2f | %.2f | %.2f | %s | %s | %s | %s | %s | %s | %s |' % (
        name, r[1], r[2] / 1e3, r[3] / 1e3, len(real), ravg / 1e3, med / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot, r[6], r[7], r[8], r[9], r[10], r[11], r[12]))
out = '\n'.join(lines)
if len(sys.argv) > 2:
    open(sys.argv[2], 'w').write(out + '\n')
print(out)

# roctx ranges (rocprofv3 --marker-trace with NVDR_ROCTX=1): host-side spans of the entry points and their stages
try:
    # the range's message travels in the event's extdata ({"message": "..."}); `name` is the API call (roctxThreadRangeA)
    regs = db.execute("""select coalesce(json_extract(extdata, '$.message'), name) as label, count(*), sum(duration), avg(duration),
                          min(duration), max(duration) from regions where category like '%MARKER%'
                          group by label order by sum(duration) desc""").fetchall()
except sqlite3.Error:
    regs = []
if regs:
    rl = ['', '| roctx range (host enqueue span) | calls | total us | avg us | min us | max us |', '|---|---|---|---|---|---|']
    for r in regs:
        rl.append('| %s | %d | %.1f | %.2f | %.2f | %.2f |' % (r[0], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3))
    rt = '\n'.join(rl)
    if len(sys.argv) > 2:
        open(sys.argv[2], 'a').write(rt + '\n')
    print(rt)
