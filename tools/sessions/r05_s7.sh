#!/bin/bash
# round 5, GPU session 7: per-kernel times of the rebuild alone (684 k and 171 k triangles, bob) with the treelet pass; the forced-schedule test
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5s7; O=$R/gpurun_out/r5s7
timeout 600 python -m pytest tests/test_gpu_distributed.py -q -k "several_rank_schedule" 2>&1 | grep -v "^$" | tail -30 | cut -c1-300 | tee $O/pytest_forced.txt
cd /tmp; export TMPDIR=/tmp
for sd in 3 2 0; do
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o r -- python $R/tools/bvh_probe.py bob $sd 30 > /tmp/kt.log 2>&1
grep triangles /tmp/kt.log
python - <<'PY' | tee -a $R/gpurun_out/r5s7/bvh_kernels.txt
import sqlite3, glob
db = sqlite3.connect(glob.glob('/tmp/kt/**/*_results.db', recursive=True)[0])
rows = db.execute('select name, count(*), avg(end - start), min(end - start) from kernels group by name order by avg(end - start) desc').fetchall()
tot = 0
for n, c, a, mn in rows:
    print('  %-70s x%-4d avg %8.1f us  min %8.1f us' % (n[:70], c, a / 1e3, mn / 1e3))
print()
PY
done
