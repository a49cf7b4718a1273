#!/bin/bash
# round-2 GPU session U (final records of the round): the whole GPU test suite + smoke, the driver's bench command with the PMC table,
# a kernel trace of the same command, the one-view iteration (HIP graph on / off), the other presets, a 2-rank dry run.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp NVDR_MESH_CACHE=/tmp
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 )) s] $*"; }
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
el "== GPU tests + smoke"
timeout 200 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-300 | tee $O/r02u_pytest.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1
el "== bench default (driver's command)"
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --pmc-keep $O/r02u_pmc > $O/r02u_bench_bob512.json 2> $O/r02u_bench.err
python tools/bench_brief.py $O/r02u_bench_bob512.json | cut -c1-1200
el "== kernel trace of the same command"
cd /tmp; rm -rf /tmp/kt
timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/kt -o r -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > /tmp/kt.log 2>&1
timeout 60 python $R/tools/rocpd_summary.py $(find /tmp/kt -name "*_results.db" | head -1) $R/$O/r02u_kernel_trace.md | cut -c1-150 | head -14
cd $R
el "== one view per iteration, HIP graph on"
timeout 90 python bench.py --batch 1 --graph on --steps 50 --warmup 10 --no-pmc --no-cpu-baseline > $O/r02u_bench_oneview_graph_on.json 2>> $O/r02u_bench.err; python tools/bench_brief.py $O/r02u_bench_oneview_graph_on.json | head -1
el "== dmtet800 (configs[3] stand-in)"
timeout 120 python bench.py --config dmtet800 --steps 10 --warmup 3 --no-extended --pmc-keep $O/r02u_pmc > $O/r02u_bench_dmtet800.json 2>> $O/r02u_bench.err; python tools/bench_brief.py $O/r02u_bench_dmtet800.json | head -5 | cut -c1-900
el "== spot512x256 (configs[2])"
timeout 90 python bench.py --config spot512x256 --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > $O/r02u_bench_spot512x256.json 2>> $O/r02u_bench.err; python tools/bench_brief.py $O/r02u_bench_spot512x256.json | head -2 | cut -c1-700
el "== 2 ranks on this one GPU (gloo collective; RCCL refuses two ranks per device)"
NVDR_BENCH_OVERSUBSCRIBE=1 NVDR_BENCH_BACKEND=gloo timeout 100 python bench.py --gpus 2 --steps 20 --warmup 5 --no-pmc --no-cpu-baseline > $O/r02u_bench_2rank_dryrun.json 2>> $O/r02u_bench.err; python tools/bench_brief.py $O/r02u_bench_2rank_dryrun.json | head -1
el "== one view per iteration, eager"
timeout 90 python bench.py --batch 1 --graph off --steps 50 --warmup 10 --no-pmc --no-cpu-baseline > $O/r02u_bench_oneview_graph_off.json 2>> $O/r02u_bench.err; python tools/bench_brief.py $O/r02u_bench_oneview_graph_off.json | head -1
tail -3 $O/r02u_bench.err | cut -c1-300; ls $O/r02u_pmc 2>/dev/null
el done
