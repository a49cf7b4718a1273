#!/bin/bash
# round 3, GPU session 47: kernel trace + iteration timeline of the dmtet800 preset (what the 10 ms of its generation stage are)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/s47; O=$R/gpurun_out/s47
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --config dmtet800 --graph off --no-cpu-baseline --no-pmc --no-extended --steps 6 --warmup 3 > /tmp/kt.log 2>&1
timeout 60 python $R/tools/rocpd_summary.py /tmp/kt/r_results.db $O/kernel_trace_dmtet800.md | cut -c1-60,100-190 | head -16
timeout 60 python $R/tools/rocpd_iteration.py /tmp/kt/r_results.db light_rows_kernel > $O/iteration_dmtet800.txt 2>&1
head -3 $O/iteration_dmtet800.txt
