#!/usr/bin/env python3
"""Per-kernel average of the PMC counters in a rocprofv3 rocpd database.  usage: rocpd_pmc.py results.db [filter]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ''
cols = [r[1] for r in db.execute('pragma table_info(pmc_events)')]
print('columns:', cols)
name_col = 'counter_name' if 'counter_name' in cols else ('name' if 'name' in cols else cols[0])
val_col = 'value' if 'value' in cols else ('counter_value' if 'counter_value' in cols else cols[-1])
try:
    q = ("select k.name, p.%s, count(*), avg(p.%s), sum(p.%s) from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id "
         "group by k.name, p.%s order by sum(p.%s) desc" % (name_col, val_col, val_col, name_col, val_col))
    rows = db.execute(q).fetchall()
except Exception as e:
    print('join failed:', e)
    rows = db.execute('select * from pmc_events limit 5').fetchall()
    print(rows)
    sys.exit(0)
print('| kernel | counter | dispatches | avg per dispatch | total |')
print('|---|---|---|---|---|')
for r in rows:
    if flt and flt not in r[0]:
        continue
    print('| %s | %s | %d | %.1f | %.1f |' % (r[0][:100], r[1], r[2], r[3], r[4]))
