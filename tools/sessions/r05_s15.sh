#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5s15
timeout 300 python bench.py --config dmtet800 --batch 1 --graph on --lock-pos on --exchange-world1 --no-cpu-baseline --no-pmc --no-large-mesh --no-other-configs --no-one-view --steps 100 --warmup 20 > gpurun_out/r5s15/out.json 2> gpurun_out/r5s15/err.log; echo rc $?; tail -c 600 gpurun_out/r5s15/out.json; grep -v "amdgpu.ids\|^$" gpurun_out/r5s15/err.log | tail -25 | cut -c1-300
