#!/bin/bash
# round 6, GPU session 21: what the traversal kernel's wavefronts wait for, by hardware counter (tools/stall_probe.py: SQ active / wait cycles by instruction class, LDS conflicts, instruction cache, TA / TD / TCP busy and stall cycles)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s21; O=$R/gpurun_out/r6s21
cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail > $O/list_avail.txt 2>&1; wc -l $O/list_avail.txt
PROBE_VIEWS=8 timeout 1500 python $R/tools/stall_probe.py $O/stall_counters_bob512_8views.md > /dev/null 2> $O/err_bob8.txt; tail -60 $O/stall_counters_bob512_8views.md
PROBE_VIEWS=1 PROBE_RES=800 PROBE_SUBDIV=3 timeout 1500 python $R/tools/stall_probe.py $O/stall_counters_684k_1view.md > /dev/null 2> $O/err_684k.txt; tail -60 $O/stall_counters_684k_1view.md
