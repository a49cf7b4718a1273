#!/bin/bash
# round 3, GPU session 32: roctx ranges under rocprofv3 (marker + kernel trace), summarised from the rocpd database
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/s32; O=$R/gpurun_out/s32
cd /tmp; export TMPDIR=/tmp
NVDR_ROCTX=1 timeout 300 rocprofv3 --marker-trace --kernel-trace -d $O/roctx -o roctx -- python $R/bench.py --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-large-mesh --no-extended > $O/roctx.log 2>&1
cd $R; db=$(find $O/roctx -name "*.db" | head -1); echo $db
python - <<P
import sqlite3
db = sqlite3.connect('$db')
print(db.execute("select category, count(*) from regions group by category").fetchall())
P
python tools/rocpd_summary.py $db $O/roctx_summary.md | tail -25
ls -la $db
