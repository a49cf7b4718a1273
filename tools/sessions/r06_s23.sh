#!/bin/bash
# round 6, GPU session 23: light-gradient gather with the records of a step added as ONE batch of compare-and-swaps (NVDR_LG_BATCH): parity, then A/B against one record at a time
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s23; O=$R/gpurun_out/r6s23
bash tools/build_variants.sh nobatch:"-DNVDR_LG_BATCH=0" 2>&1 | tail -1
echo "== tests"; timeout 1500 python -m pytest tests/test_gpu_env_shade.py tests/test_gpu_fullsize.py tests/test_gpu_geometry.py -q -m gpu -x 2>&1 | grep -v Warning | tail -3 | tee $O/tests.txt
ab() { out=$1; shift; env "$@" timeout 900 python tools/ab_inproc.py ${ROUNDS:-5} 2>&1 | grep -v "Warning\|tenancy\|amdgpu.ids" | head -5 | tee $O/$out.txt; }
ab ab_bob1 PROBE_VIEWS=1
ab ab_bob8 PROBE_VIEWS=8
ab ab_684k_1 PROBE_VIEWS=1 PROBE_RES=800 PROBE_SUBDIV=3
ROUNDS=3 ab ab_684k_8 PROBE_VIEWS=8 PROBE_RES=800 PROBE_SUBDIV=3
cd /tmp; export TMPDIR=/tmp
for v in current nobatch; do
  for views in 1 8; do
  rm -rf /tmp/kt; AB_ONLY=$v AB_ITERS=6 PROBE_VIEWS=$views timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/tools/ab_inproc.py 1 > /tmp/kt.log 2>&1
  echo "== $v, $views view(s) (kernel trace; a variant's rows average it with the current library's launches: 14 + 14 calls)"; timeout 60 python $R/tools/rocpd_summary.py /tmp/kt/r_results.db /tmp/sum.md | grep "light_grad_block" | cut -d'|' -f2-8 | cut -c1-160
  done
done 2>&1 | tee $O/lg_kernel_trace.txt
