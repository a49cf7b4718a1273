#!/bin/bash
# round-2 GPU session F: the record runs -- bench default (as the driver runs it) with the PMC counter table, kernel trace of the same
# command, the other presets, the one-view iteration eager vs HIP graph, a 2-rank dry run.  Small text / JSON outputs only.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 300 python -m pytest tests/test_gpu_env_shade.py tests/test_gpu_fullsize.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-200
echo "== bench default (driver's command)"
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --pmc-keep $O/r02f_pmc > $O/r02f_bench_bob512.json 2> $O/r02f_bench.err
python tools/bench_brief.py $O/r02f_bench_bob512.json | cut -c1-900
echo "== kernel trace of the same command"
cd /tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o r -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > /tmp/kt.log 2>&1
timeout 60 python $R/tools/rocpd_summary.py $(find /tmp/kt -name "*_results.db" | head -1) $R/$O/r02f_kernel_trace.md | cut -c1-150 | head -12
cd $R
echo "== one view per iteration: eager vs HIP graph"
for g in off on; do timeout 300 python bench.py --batch 1 --graph $g --steps 50 --warmup 10 --no-pmc --no-cpu-baseline > $O/r02f_bench_oneview_graph_$g.json 2>> $O/r02f_bench.err; python tools/bench_brief.py $O/r02f_bench_oneview_graph_$g.json | head -1; done
echo "== 2 ranks on this one GPU (gloo collective; RCCL refuses two ranks per device)"
NVDR_BENCH_OVERSUBSCRIBE=1 NVDR_BENCH_BACKEND=gloo timeout 400 python bench.py --gpus 2 --steps 20 --warmup 5 --no-pmc --no-cpu-baseline > $O/r02f_bench_2rank_dryrun.json 2>> $O/r02f_bench.err; python tools/bench_brief.py $O/r02f_bench_2rank_dryrun.json | head -1
echo "== spot512x256 (configs[2])"
timeout 400 python bench.py --config spot512x256 --steps 10 --warmup 3 --no-cpu-baseline --pmc-keep $O/r02f_pmc > $O/r02f_bench_spot512x256.json 2>> $O/r02f_bench.err; python tools/bench_brief.py $O/r02f_bench_spot512x256.json | head -4 | cut -c1-700
echo "== dmtet800 (configs[3] stand-in)"
timeout 500 python bench.py --config dmtet800 --steps 10 --warmup 3 --no-extended --pmc-keep $O/r02f_pmc > $O/r02f_bench_dmtet800.json 2>> $O/r02f_bench.err; python tools/bench_brief.py $O/r02f_bench_dmtet800.json | head -5 | cut -c1-700
tail -3 $O/r02f_bench.err | cut -c1-300; ls -la $O/r02f_pmc
