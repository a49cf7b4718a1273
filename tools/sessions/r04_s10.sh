#!/bin/bash
# round 4, GPU session 10: the dependency-free eight-wide builder + the one-launch bounds kernel: BVH tests, then build timings
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4s10; O=$R/gpurun_out/r4s10
timeout 900 python -m pytest tests/test_gpu_bvh.py -x -q 2>&1 | tail -8
B="--no-cpu-baseline --no-pmc --no-large-mesh --steps 30 --warmup 10 --no-extended"
cd /tmp; export TMPDIR=/tmp
for cfg in dmtet800 bob512; do
rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --config $cfg --batch 1 --graph off --lock-pos on $B > /tmp/kt.log 2>&1
tail -1 /tmp/kt.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg 1 view eager under rocprof', d['ms_per_step'])"
timeout 60 python $R/tools/rocpd_summary.py /tmp/kt/r_results.db $O/kernel_trace_oneview_$cfg.md | cut -d'|' -f2-8,17,18 | cut -c1-160 | grep -i "bvh\|rocprim\|env_trace\|env_gen\|kernel |"
done
