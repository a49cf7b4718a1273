#!/bin/bash
# round 6, GPU session 5: split walks in the drain (parity, then A/B against -DNVDR_TRACE_SPLIT=0); is the traversal bound by the L1's request rate?
# (A/B: the node's four 16-byte requests issued twice); the one-view schedule run that produced no line in session 4
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s5; O=$R/gpurun_out/r6s5
timeout 1500 python -m pytest tests/test_gpu_bvh.py tests/test_gpu_env_shade.py tests/test_gpu_fullsize.py -q -m gpu -x 2>&1 | grep -v Warning | tail -6 | tee $O/tests.txt
bash tools/build_variants.sh nosplit:"-DNVDR_TRACE_SPLIT=0" split16:"-DNVDR_TRACE_SPLIT_FREE=16" split32e2:"-DNVDR_TRACE_SPLIT_FREE=32 -DNVDR_TRACE_SPLIT_EVERY=2" dup2:"-DNVDR_TRACE_DUP_FETCH=2" dup4:"-DNVDR_TRACE_DUP_FETCH=4" 2>&1 | tail -5
PROBE_VIEWS=1 timeout 600 python tools/ab_inproc.py 5 2>&1 | grep -v "Warning\|tenancy\|^  [a-z0-9]* *[0-9.]* [0-9.]* [0-9.]*" | tee $O/ab_bob1.txt
PROBE_VIEWS=8 timeout 600 python tools/ab_inproc.py 4 2>&1 | grep -v "Warning\|tenancy\|^  [a-z0-9]* *[0-9.]* [0-9.]* [0-9.]*" | tee $O/ab_bob8.txt
PROBE_VIEWS=1 PROBE_RES=800 PROBE_SUBDIV=3 AB_ONLY=nosplit,dup4 timeout 600 python tools/ab_inproc.py 4 2>&1 | grep -v "Warning\|tenancy\|^  [a-z0-9]* *[0-9.]* [0-9.]* [0-9.]*" | tee $O/ab_dmtet800_1.txt
PROBE_VIEWS=8 PROBE_RES=800 PROBE_SUBDIV=3 AB_ONLY=nosplit,dup4 timeout 600 python tools/ab_inproc.py 3 2>&1 | grep -v "Warning\|tenancy\|^  [a-z0-9]* *[0-9.]* [0-9.]* [0-9.]*" | tee $O/ab_dmtet800_8.txt
timeout 200 python -X faulthandler -c "
import faulthandler, sys, runpy
faulthandler.dump_traceback_later(120, exit=True)
sys.argv = ['bench.py', '--config', 'dmtet800', '--batch', '1', '--graph', 'on', '--exchange-world1', '--no-cpu-baseline', '--no-pmc', '--no-large-mesh', '--no-other-configs', '--no-one-view', '--no-validation', '--steps', '60', '--warmup', '10']
runpy.run_path('bench.py', run_name='__main__')
" 2>&1 | grep -v "Warning\|RCCL\|HIP version\|ROCm\|Hostname\|Librccl" | tail -30 | cut -c1-400
