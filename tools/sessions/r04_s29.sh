#!/bin/bash
# round 4, GPU session 29: calibration of the memory-side counters on random 64-byte gathers (tools/ubench/gather64.hip)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4s29; O=$R/gpurun_out/r4s29
hipcc --offload-arch=gfx950 -O3 tools/ubench/gather64.hip -o /tmp/gather64 2>/dev/null
/tmp/gather64 | tee $O/gather64_times.txt
cd /tmp; export TMPDIR=/tmp
for grp in "FETCH_SIZE" "TCC_REQ_sum TCC_MISS_sum WRITE_SIZE" "TCC_HIT_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
rm -rf /tmp/pm
timeout 120 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pm -o r -- /tmp/gather64 > /tmp/pm.log 2>&1
python3 $R/tools/rocpd_pmc.py /tmp/pm/r_results.db gather64 | grep "gather64" | tee -a $O/gather64_pmc.txt
done
