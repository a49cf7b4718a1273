#!/bin/bash
# round 5, GPU session 22: is the traversal kernel waiting for its node fetches?  vector-memory latency (SQ_INST_LEVEL_VMEM / SQ_INSTS_VMEM_RD), L1 (TCP) and
# texture-addresser counters of env_trace_kernel<false>, 8 views of bob and of the 684 k-triangle mesh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5s22; O=$R/gpurun_out/r5s22
export TMPDIR=/tmp
run() {  # tag config
  i=0
  for G in "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TD_TD_BUSY_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SALU"; do
    i=$((i+1)); rm -rf /tmp/pmc_$1_$i
    ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $G -d /tmp/pmc_$1_$i -o r -- python $R/bench.py --pmc-child --config $2 --steps 2 --warmup 1 > /tmp/pmc_$1_$i.log 2>&1 ) || echo "group $i failed: $(tail -3 /tmp/pmc_$1_$i.log)" >> $O/$1.md
    db=$(find /tmp/pmc_$1_$i -name '*_results.db' | head -1)
    [ -n "$db" ] && timeout 60 python tools/rocpd_pmc.py $db env_trace_kernel | grep -v "^columns" >> $O/$1.md 2>&1
  done
  cut -c1-40,120-200 $O/$1.md
}
run bob8 bob512
run dmtet8 dmtet800
