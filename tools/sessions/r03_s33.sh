#!/bin/bash
# round 3, GPU session 33: timeline of one 8-view and one 1-view iteration (is the side-stream BVH build on the critical path?)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/s33; O=$R/gpurun_out/s33
cd /tmp; export TMPDIR=/tmp
for b in 8 1; do
rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --batch $b --graph off --no-cpu-baseline --no-pmc --no-extended --no-large-mesh --steps 10 --warmup 5 > /tmp/kt.log 2>&1
timeout 60 python $R/tools/rocpd_iteration.py /tmp/kt/r_results.db light_rows_kernel > $O/iteration_b$b.txt 2>&1
head -3 $O/iteration_b$b.txt
done
