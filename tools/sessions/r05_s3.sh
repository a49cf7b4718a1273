#!/bin/bash
# round 5, GPU session 3: the forced-schedule test again; how sparse the texture gradient is at which tile size; graph-mode timelines of the one-view iteration
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5s3; O=$R/gpurun_out/r5s3
timeout 600 python -m pytest tests/test_gpu_distributed.py -q -k "several_rank_schedule or rccl_world1" 2>&1 | tail -3
for p in bob512 dmtet800 spot512x256; do timeout 200 python tools/tile_fraction_probe.py $p 2>&1 | grep -v Warning; done | tee $O/tile_fractions.txt
cd /tmp; export TMPDIR=/tmp
trace() { name=$1; shift; rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py "$@" --no-cpu-baseline --no-pmc --no-extended --no-large-mesh --no-other-configs --no-one-view > /tmp/kt.log 2>&1
  timeout 60 python $R/tools/rocpd_iteration.py /tmp/kt/r_results.db light_rows_kernel -8 > $O/${name}_iteration.txt 2>&1; head -1 $O/${name}_iteration.txt; tail -3 /tmp/kt.log | cut -c1-300; }
trace kt_bob1_graph --batch 1 --graph on --steps 40 --warmup 10
trace kt_bob1_graph_xsparse --batch 1 --graph on --steps 40 --warmup 10 --exchange-world1
trace kt_bob1_graph_xdense --batch 1 --graph on --steps 40 --warmup 10 --exchange-world1 --exchange dense
