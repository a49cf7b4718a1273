#!/bin/bash
# round 5, GPU session 26: is the traversal kernel bound by the L1 (TCP / TA) rather than by instruction issue?  the counters this rocprofv3 accepts, then one pass each
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5s26; O=$R/gpurun_out/r5s26
export TMPDIR=/tmp
( cd /tmp && timeout 60 rocprofv3 -L 2>/dev/null | grep -o -E "\b(TCP|TA|TD|TCC|SQ_INSTS_VMEM|SQ_INST_LEVEL|SQ_WAIT)_[A-Za-z0-9_]+" | sort -u > $O/counters_available.txt ); wc -l $O/counters_available.txt
run() {  # tag counters...
  tag=$1; shift; rm -rf /tmp/pmc_$tag
  ( cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmc_$tag -o r -- python $R/bench.py --pmc-child --config bob512 --steps 2 --warmup 1 > /tmp/pmc_$tag.log 2>&1 ) || { echo "$tag failed: $(grep -i -m3 'error\|invalid\|not' /tmp/pmc_$tag.log | cut -c1-200)" | tee -a $O/bob8.md; return; }
  db=$(find /tmp/pmc_$tag -name '*_results.db' | head -1)
  [ -n "$db" ] && timeout 60 python tools/rocpd_pmc.py $db env_trace_kernel | grep -v "^columns" | tee -a $O/bob8.md | cut -c1-40,120-200
}
has() { grep -q -x "$1" $O/counters_available.txt; }
G1=""; for c in TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum; do has $c && G1="$G1 $c"; done
G2=""; for c in TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum TD_TD_BUSY_sum; do has $c && G2="$G2 $c"; done
echo "G1:$G1"; echo "G2:$G2"
[ -n "$G1" ] && run g1 $G1
[ -n "$G2" ] && run g2 $G2
