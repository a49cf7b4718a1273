#!/bin/bash
# round 4, GPU session 18: timeline of ONE one-view iteration (bob512 locked, dmtet800 unlocked): the launches to fold
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4s18; O=$R/gpurun_out/r4s18
cd /tmp; export TMPDIR=/tmp
for cfg in bob512 dmtet800; do
rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --config $cfg --batch 1 --graph off --no-cpu-baseline --no-pmc --no-extended --no-large-mesh --steps 10 --warmup 5 > /tmp/kt.log 2>&1
timeout 60 python $R/tools/rocpd_iteration.py /tmp/kt/r_results.db light_rows_kernel > $O/iteration_1view_$cfg.txt 2>&1
cut -c1-170 $O/iteration_1view_$cfg.txt
done
