#!/bin/bash
# usage: pmc_run.sh <tag> <kernel-filter> -- <command...>   : runs the command under several rocprofv3 PMC passes (each under
# `timeout`) and writes per-kernel counter averages to gpurun_out/<tag>_pmc.md (one run per counter group; no tracing domains mixed in)
TAG=$1; FILTER=$2; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/${TAG}_pmc.md; : > $OUT
i=0
for G in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU" \
         "SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_LATENCY_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 120 rocprofv3 --kernel-trace --pmc $G -d /tmp/pmc_$i -o r -- "$@" > /tmp/pmc_$i.log 2>&1 || echo "group $i failed/timeout" >> $OUT
  timeout 60 python $R/tools/rocpd_pmc.py /tmp/pmc_$i/r_results.db "$FILTER" | grep -v "^columns" >> $OUT 2>&1
done
cut -c1-190 $OUT
