#!/bin/bash
# round 6, GPU session 17: the one-view light-gradient gather is ~60 us even with nothing to do but scan its tags: is it the 96 KB of LDS / the 1024-thread workgroups?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s17; O=$R/gpurun_out/r6s17
bash tools/build_variants.sh lgt256:"-DNVDR_LG_THREADS=256" lgt512:"-DNVDR_LG_THREADS=512" 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp
run() { tag=$1; shift; rm -rf /tmp/kt; env "$@" AB_ITERS=6 PROBE_VIEWS=${VIEWS:-1} timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/tools/ab_inproc.py 2 > /tmp/kt.log 2>&1
  echo "== $tag"; timeout 60 python $R/tools/rocpd_summary.py /tmp/kt/r_results.db /tmp/sum.md | grep "light_grad" | cut -d'|' -f2-8 | cut -c1-160; }
( run lds96_default AB_ONLY=current
  run lds24 AB_ONLY=x AB_ENV="x:NVDR_LG_LDS_KB=24"
  run lds48 AB_ONLY=x AB_ENV="x:NVDR_LG_LDS_KB=48"
  run lds64 AB_ONLY=x AB_ENV="x:NVDR_LG_LDS_KB=64"
  run lds128 AB_ONLY=x AB_ENV="x:NVDR_LG_LDS_KB=128"
  run allbands AB_ONLY=x AB_ENV="x:NVDR_LG_MODE=0"
  run threads256 AB_ONLY=lgt256
  run threads512 AB_ONLY=lgt512
  run threads256_lds48 AB_ONLY=x2 AB_ENV="x2:NVDR_LG_LDS_KB=48" ) 2>&1 | tee $O/lg_lds_threads_1view.txt
