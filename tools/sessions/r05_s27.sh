#!/bin/bash
# round 5, GPU session 27: L1 (TCP) / texture-addresser (TA) counters of the traversal kernel, two counters per block and pass (a larger group is refused: "exceeds the capabilities of the hardware")
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5s27; O=$R/gpurun_out/r5s27
export TMPDIR=/tmp
run() {  # tag counters...
  tag=$1; shift; rm -rf /tmp/pmc_$tag
  ( cd /tmp && timeout 45 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmc_$tag -o r -- python $R/bench.py --pmc-child --config bob512 --steps 2 --warmup 1 > /tmp/pmc_$tag.log 2>&1 ) || { echo "$tag ($*) failed: $(grep -i -m2 'error code\|invalid' /tmp/pmc_$tag.log | cut -c1-160)" | tee -a $O/bob8.md; return; }
  db=$(find /tmp/pmc_$tag -name '*_results.db' | head -1)
  [ -n "$db" ] && timeout 60 python tools/rocpd_pmc.py $db env_trace_kernel | grep -v "^columns\|^|---\|^| kernel" | tee -a $O/bob8.md | awk -F'|' '{print $3, $4, $5}'
}
run a TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
run b TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum
run c TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run d TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run e TD_TD_BUSY_sum TCP_TA_TCP_STATE_READ_sum
