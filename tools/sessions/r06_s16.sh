#!/bin/bash
# round 6, GPU session 16: what the one-view light-gradient gather consists of (timing-only variants: no adds / no record loads / no partial rows / no zeroing),
# and more eager split-walk settings
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s16; O=$R/gpurun_out/r6s16
bash tools/build_variants.sh noadd:"-DNVDR_LG_SKIP=1" noload:"-DNVDR_LG_SKIP=2" norow:"-DNVDR_LG_SKIP=4" nozero:"-DNVDR_LG_SKIP=8" nothing:"-DNVDR_LG_SKIP=15" sp4e1:"-DNVDR_TRACE_SPLIT_FREE=4" sp16e1:"-DNVDR_TRACE_SPLIT_FREE=16" sp1e1:"-DNVDR_TRACE_SPLIT_FREE=1" 2>&1 | tail -8
cd /tmp; export TMPDIR=/tmp
for v in current noadd noload norow nozero nothing; do
  rm -rf /tmp/kt; AB_ONLY=$v AB_ITERS=6 PROBE_VIEWS=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/tools/ab_inproc.py 2 > /tmp/kt.log 2>&1
  echo "== $v"; timeout 60 python $R/tools/rocpd_summary.py /tmp/kt/r_results.db /tmp/sum.md | grep "light_grad\|env_shade_queue_kernel<true" | cut -d'|' -f2-8 | cut -c1-160
done 2>&1 | tee $O/lg_variants_1view.txt
cd $R
NVDR_TUNING=1 NVDR_TRACE_SPLIT_MODE=1 PROBE_VIEWS=1 AB_ONLY=sp4e1,sp16e1,sp1e1 timeout 600 python tools/ab_inproc.py 5 2>&1 | grep -v "Warning\|tenancy\|amdgpu.ids" | head -7 | tee $O/ab_split_bob1.txt
NVDR_TUNING=1 NVDR_TRACE_SPLIT_MODE=1 PROBE_VIEWS=1 PROBE_RES=800 PROBE_SUBDIV=3 AB_ONLY=sp4e1,sp16e1,sp1e1 timeout 600 python tools/ab_inproc.py 4 2>&1 | grep -v "Warning\|tenancy\|amdgpu.ids" | head -7 | tee $O/ab_split_dmtet800_1.txt
