#!/bin/bash
# round 6, GPU session 24: light-gradient gather with fp64 accumulators and ds_add_f64 (where the probe then fits 16 bands): parity, A/B against the fp32 compare-and-swap build (NVDR_LG_F64=0)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s24; O=$R/gpurun_out/r6s24
echo "== tests"; timeout 1500 python -m pytest tests/test_gpu_env_shade.py tests/test_gpu_fullsize.py tests/test_gpu_geometry.py tests/test_gpu_distributed.py -q -m gpu -x 2>&1 | grep -v Warning | tail -3 | tee $O/tests.txt
E="f32:NVDR_LG_F64=0"
ab() { out=$1; shift; env "$@" AB_ENV="$E" timeout 900 python tools/ab_inproc.py ${ROUNDS:-5} 2>&1 | grep -v "Warning\|tenancy\|amdgpu.ids" | head -5 | tee $O/$out.txt; }
ab ab_bob1 PROBE_VIEWS=1
ab ab_bob8 PROBE_VIEWS=8
ab ab_684k_1 PROBE_VIEWS=1 PROBE_RES=800 PROBE_SUBDIV=3
ROUNDS=3 ab ab_684k_8 PROBE_VIEWS=8 PROBE_RES=800 PROBE_SUBDIV=3
cd /tmp; export TMPDIR=/tmp
for v in 1 0; do
  for views in 1 8; do
  rm -rf /tmp/kt; NVDR_TUNING=1 NVDR_LG_F64=$v AB_ONLY=current AB_ITERS=6 PROBE_VIEWS=$views timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/tools/ab_inproc.py 1 > /tmp/kt.log 2>&1
  echo "== NVDR_LG_F64=$v, $views view(s)"; timeout 60 python $R/tools/rocpd_summary.py /tmp/kt/r_results.db /tmp/sum.md | grep "light_grad_\|env_shade_queue_kernel<true" | cut -d'|' -f2-8 | cut -c1-150
  done
done 2>&1 | tee $O/lg_kernel_trace.txt
